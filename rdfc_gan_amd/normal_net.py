"""G_normal's decoder on the HIP kernels.

Upstream's surface-normal net is an EfficientNet-B5 encoder fetched through torch.hub
(lib/models/generator/normal_net/submodules/encoder.py:7-31 -- third-party code, not in the tree, no network here) and the
in-tree decoder (submodules/decoder.py:6-25, UpSampleBN submodules.py:10-24).  This module rebuilds the decoder and the
NNET wrapper (NNET.py:9-25) with upstream's state_dict keys; the encoder is whatever nn.Module the caller passes
(stock PyTorch on the same device), called exactly as upstream calls it: it returns the feature list and the decoder
takes entries 4, 5, 6, 8 and 11.  Gradients flow back into the encoder's features.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from .layers import run_conv


class UpSampleBN(nn.Module):
    def __init__(self, skip_input, output_features):
        super().__init__()
        self._net = nn.Sequential(nn.Conv2d(skip_input, output_features, 3, 1, 1), nn.BatchNorm2d(output_features), nn.LeakyReLU(),
                                  nn.Conv2d(output_features, output_features, 3, 1, 1), nn.BatchNorm2d(output_features),
                                  nn.LeakyReLU())

    def forward(self, x, concat_with):
        """pixel-major maps: bilinear (align_corners) resize of x to the skip's size, channel concat, two conv+BN+LeakyReLU"""
        f = torch.cat([Fh.interpolate_bilinear(x, tuple(concat_with.shape[1:3])), concat_with], 3)
        n = self._net
        return run_conv(run_conv(f, n[0], n[1], n[2]), n[3], n[4], n[5])


class Decoder(nn.Module):
    def __init__(self, num_classes=4):
        super().__init__()
        self.conv2 = nn.Conv2d(2048, 2048, 1)
        self.up1 = UpSampleBN(2048 + 176, 1024)
        self.up2 = UpSampleBN(1024 + 64, 512)
        self.up3 = UpSampleBN(512 + 40, 256)
        self.up4 = UpSampleBN(256 + 24, 128)
        self.conv3 = nn.Conv2d(128, num_classes, 3, 1, 1)

    def forward_pm(self, features):
        b0, b1, b2, b3, b4 = (Fh.to_pm(features[i]) for i in (4, 5, 6, 8, 11))
        x = run_conv(b4, self.conv2, None, None)
        x = self.up4(self.up3(self.up2(self.up1(x, b3), b2), b1), b0)
        return run_conv(x, self.conv3, None, None, out_f32=True)          # [N, h, w, classes] f32

    def forward(self, features):
        return self.forward_pm(features).permute(0, 3, 1, 2)              # upstream's [N, classes, h, w]


class NNET(nn.Module):
    def __init__(self, out_height, out_width, encoder=None):
        super().__init__()
        if encoder is None:
            raise ValueError('NNET needs an encoder module: upstream downloads tf_efficientnet_b5_ap through torch.hub '
                             '(normal_net/submodules/encoder.py:12); pass that model (or any module returning its feature list)')
        self.encoder = encoder
        self.decoder = Decoder()
        self.out_height, self.out_width = out_height, out_width

    def get_1x_lr_params(self):
        return self.encoder.parameters()

    def get_10x_lr_params(self):
        return self.decoder.parameters()

    def forward(self, img):
        out = self.decoder.forward_pm(self.encoder(img))
        return Fh.interpolate_bilinear(out, (self.out_height, self.out_width)).permute(0, 3, 1, 2)

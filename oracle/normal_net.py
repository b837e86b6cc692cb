"""CPU oracle (test infrastructure only) of G_normal's in-tree part: the decoder and the NNET wrapper.

Plain torch.nn restatement, NCHW f32, upstream's state_dict keys:
  Decoder      lib/models/generator/normal_net/submodules/decoder.py:6-25
  UpSampleBN   lib/models/generator/normal_net/submodules/submodules.py:10-24
  NNET         lib/models/generator/normal_net/NNET.py:9-25 (the torch.hub EfficientNet encoder, encoder.py:7-31, is third-party
               code that is not in the tree: the encoder is a constructor argument here)
Pinned by tests/golden/normal_decoder_small.npz (upstream's own Decoder class, tests/golden/make_golden.py::normal_decoder).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class UpSampleBN(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self._net = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.LeakyReLU(),
                                  nn.Conv2d(cout, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.LeakyReLU())

    def forward(self, x, skip):
        up = F.interpolate(x, size=skip.shape[2:], mode='bilinear', align_corners=True)
        return self._net(torch.cat([up, skip], 1))


class Decoder(nn.Module):
    def __init__(self, num_classes=4):
        super().__init__()
        self.conv2 = nn.Conv2d(2048, 2048, 1)
        self.up1, self.up2 = UpSampleBN(2048 + 176, 1024), UpSampleBN(1024 + 64, 512)
        self.up3, self.up4 = UpSampleBN(512 + 40, 256), UpSampleBN(256 + 24, 128)
        self.conv3 = nn.Conv2d(128, num_classes, 3, 1, 1)

    def forward(self, features):
        b0, b1, b2, b3, b4 = (features[i] for i in (4, 5, 6, 8, 11))
        x = self.up1(self.conv2(b4), b3)
        return self.conv3(self.up4(self.up3(self.up2(x, b2), b1), b0))


class NNET(nn.Module):
    def __init__(self, out_height, out_width, encoder):
        super().__init__()
        self.encoder, self.decoder = encoder, Decoder()
        self.out_height, self.out_width = out_height, out_width

    def forward(self, img):
        out = self.decoder(self.encoder(img))
        return F.interpolate(out, size=[self.out_height, self.out_width], mode='bilinear', align_corners=True)

"""Import the upstream RDFC-GAN python under scratch stubs (build container only).

Used ONLY by make_golden.py to produce the committed fixtures; nothing here runs
on the GPU box (``/root/reference`` does not exist there).  The stubs stand in
for modules the image lacks or that upstream never shipped; none of them is on
the arithmetic path that the fixtures pin, except where stated:

* ``DCN``                    upstream's CUDA-only extension -> the C oracle
                             (so NLSPN fixtures pin upstream's *python* NLSPN logic
                             around the op; the op itself is pinned by properties)
* ``torchvision``            resnet18/34 BasicBlock stack with torchvision's
                             attribute names (plain torch.nn arithmetic)
* ``lib.models.generator.build_generator`` / ``...discriminator.build_discriminator``
  / ``lib.utils.dist_utils`` git-ignored upstream -> 3-line registries
* ``NNET`` / ``PSPNet``      out-of-scope nets -> constant-output stand-ins
* cv2, imageio, open3d, ...  import-only dummies

One behavioural patch: torch.nn.functional.instance_norm is wrapped so that the cotangent
reaching its backward is made contiguous.  PyTorch 2.10's CPU instance_norm backward returns
wrong input gradients for a channels-last-strided cotangent (batch size 1 behind a
ConvTranspose2d); the wrapper only removes that stride pattern -- values, and the gradient
upstream computes on a correct PyTorch, are unchanged (checked against the closed form).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference/RDFC-GAN'
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Dummy(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        sub = _Dummy(self.__name__ + '.' + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return self


def _resnet(layers):
    class Block(nn.Module):
        expansion = 1

        def __init__(self, cin, cout, stride=1, downsample=None):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)
            self.downsample = downsample

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idt)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            cin = 64
            for i, (c, n) in enumerate(zip((64, 128, 256, 512), layers), 1):
                blocks = []
                for j in range(n):
                    s = 2 if (j == 0 and i > 1) else 1
                    ds = None
                    if s != 1 or cin != c:
                        ds = nn.Sequential(nn.Conv2d(cin, c, 1, s, bias=False), nn.BatchNorm2d(c))
                    blocks.append(Block(cin, c, s, ds))
                    cin = c
                setattr(self, f'layer{i}', nn.Sequential(*blocks))
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(512, 1000)

    return Net


class _ContiguousCotangent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()          # a copy: upstream applies in-place activations to the result

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _patch_instance_norm():
    import torch.nn.functional as F
    if getattr(F.instance_norm, '_rdfc_patched', False):
        return
    orig = F.instance_norm

    def instance_norm(*a, **k):
        return _ContiguousCotangent.apply(orig(*a, **k))
    instance_norm._rdfc_patched = True
    F.instance_norm = instance_norm


def install():
    if 'lib.models.generator.rdf_generator.rdf_generator' in sys.modules:
        return
    _patch_instance_norm()
    if not os.path.isdir(REF):
        raise RuntimeError('reference tree not present (ref_harness is build-container only)')
    for p in (REPO, os.path.join(REF, 'lib'), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    for name in ('cv2', 'imageio', 'open3d', 'h5py', 'numba', 'skimage', 'attrdict', 'tensorboardX'):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = _Dummy(name)
    # torchvision
    tv = _Dummy('torchvision')
    tvm = _Dummy('torchvision.models')
    tvm.resnet18 = lambda pretrained=False: _resnet((2, 2, 2, 2))()
    tvm.resnet34 = lambda pretrained=False: _resnet((3, 4, 6, 3))()
    tv.models = tvm
    tv.transforms = _Dummy('torchvision.transforms')
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.transforms': tv.transforms})
    # DCN -> C oracle
    from oracle import mdconv as om
    dcn = types.ModuleType('DCN')

    def _fwd(inp, w, b, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, g, dg, step):
        return om.forward(inp, off, msk, w, b, (sh, sw), (ph, pw), (dh, dw), g, dg)

    def _bwd(inp, w, b, off, msk, go, kh, kw, sh, sw, ph, pw, dh, dw, g, dg, step):
        gi, goff, gm, gw, gb = om.backward(inp, off, msk, w, go, (sh, sw), (ph, pw), (dh, dw), g, dg)
        return [gi, goff, gm, gw, gb]

    dcn.modulated_deform_conv_forward, dcn.modulated_deform_conv_backward = _fwd, _bwd
    sys.modules['DCN'] = dcn
    # git-ignored upstream files
    du = types.ModuleType('lib.utils.dist_utils')
    du.get_dist_info = lambda: (0, 1)
    sys.modules['lib.utils.dist_utils'] = du

    def _kwargs(node):
        d = dict(node)
        d.pop('type')
        return d

    bg = types.ModuleType('lib.models.generator.build_generator')

    def build_generator(node):
        from lib.models.generator.rdf_generator.rdf_generator import RDFGenerator
        from lib.models.generator.resnet_generator import ResnetGenerator
        return dict(RDFGenerator=RDFGenerator, ResnetGenerator=ResnetGenerator)[node['type']](**_kwargs(node))

    bg.build_generator = build_generator
    sys.modules['lib.models.generator.build_generator'] = bg
    bd = types.ModuleType('lib.models.discriminator.build_discriminator')

    def build_discriminator(node):
        from lib.models.discriminator.patch_gan_discriminator import PatchGANDiscriminator
        return PatchGANDiscriminator(**_kwargs(node))

    bd.build_discriminator = build_discriminator
    sys.modules['lib.models.discriminator.build_discriminator'] = bd

    # out-of-scope nets: constant outputs (no dependence on the hot-path nets)
    class NNET(nn.Module):
        queue = []           # tensors (B,4,H,W) popped per call

        def __init__(self, *a, **k):
            super().__init__()
            self.dummy = nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return NNET.queue.pop(0)

    class PSPNet(nn.Module):
        def __init__(self, num_classes):
            super().__init__()
            self.nc = num_classes
            self.dummy = nn.Parameter(torch.zeros(1))

        def forward(self, x):
            B, _, H, W = x.shape
            z = torch.zeros(B, self.nc, H, W) + 0.0 * self.dummy
            return [z, z]

    nn_mod = types.ModuleType('lib.models.generator.normal_net.NNET')
    nn_mod.NNET = NNET
    sys.modules['lib.models.generator.normal_net.NNET'] = nn_mod
    ps_mod = types.ModuleType('lib.models.generator.label_net.labels_model')
    ps_mod.PSPNet = PSPNet
    sys.modules['lib.models.generator.label_net.labels_model'] = ps_mod


def ref_classes():
    install()
    from lib.models.generator.rdf_generator.rdf_generator import RDFGenerator
    from lib.models.generator.resnet_generator import ResnetGenerator
    from lib.models.discriminator.patch_gan_discriminator import PatchGANDiscriminator
    from lib.losses import gan_loss
    return RDFGenerator, ResnetGenerator, PatchGANDiscriminator, gan_loss


def real_pspnet():
    """upstream's own PSPNet class (install() leaves a constant-output stand-in under its module name for the step trace;
    the label-net fixtures need the real one: lib/models/generator/label_net/labels_model.py:43-94)"""
    install()
    name = 'lib.models.generator.label_net.labels_model'
    stub = sys.modules.pop(name)
    try:
        real = importlib.import_module(name)
    finally:
        sys.modules[name] = stub
    return real.PSPNet


def ref_nnet_with_stand_in_encoder():
    """class with NNET's constructor signature (NNET.py:10-15) built from upstream's own Decoder and the stand-in encoder of
    cases.py; forward as NNET.py:21-25"""
    install()
    import torch.nn.functional as F
    Dec = importlib.import_module('lib.models.generator.normal_net.submodules.decoder').Decoder
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cases import StandInEncoder

    class RefNNET(nn.Module):
        def __init__(self, out_height, out_width):
            super().__init__()
            self.encoder, self.decoder = StandInEncoder(), Dec()
            self.out_height, self.out_width = out_height, out_width

        def forward(self, img):
            out = self.decoder(self.encoder(img))
            return F.interpolate(out, size=[self.out_height, self.out_width], mode='bilinear', align_corners=True)

    return RefNNET

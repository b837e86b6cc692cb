"""HIP-backed generators with upstream's constructor / forward signatures and state_dict keys.

RDFGenerator     <- RDFC-GAN/lib/models/generator/rdf_generator/rdf_generator.py:32-414
EncoderDecoder   <- .../encoder_decoder/encoder_decoder.py:5-122
AdaptiveInstanceNorm <- .../model_utils.py:53-90 (EqualLR :7-36 keeps `weight_orig`)
ResnetGenerator  <- RDFC-GAN/lib/models/generator/resnet_generator.py:6-100
build_generator  <- the git-ignored lib/models/generator/build_generator.py (call site rdfc_gan.py:81-82)

Inputs / outputs are upstream's NCHW f32 tensors; inside, maps are pixel-major in the
compute dtype (functional.set_compute_dtype).
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import functional as Fh
from .layers import ConvSeq, conv_bn_relu, convt_bn_relu, resnet_stage, run_conv
from .nlspn import NLSPNRefineModule

_RESNET_LAYERS = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3)}


class EncoderDecoder(nn.Module):
    def __init__(self, encoder_type='resnet34', skip_type='concat', encoder_channels=(64, 128, 256, 512, 512),
                 decoder_channels=(256, 128, 64, 64), pretrained_on_imagenet=False):
        super().__init__()
        if encoder_type not in _RESNET_LAYERS:
            raise NotImplementedError
        if pretrained_on_imagenet:
            raise NotImplementedError('ImageNet weights are not part of this package; load a checkpoint instead '
                                      '(upstream overwrites them in init_weights anyway, rdfc_gan.py:120)')
        nb = _RESNET_LAYERS[encoder_type]
        e, d = list(encoder_channels), list(decoder_channels)
        cat = skip_type == 'concat'
        self.en2 = resnet_stage(64, 64, 1, nb[0])
        self.en3 = resnet_stage(64, 128, 2, nb[1])
        self.en4 = resnet_stage(128, 256, 2, nb[2])
        self.en5 = resnet_stage(256, 512, 2, nb[3])
        self.en6 = conv_bn_relu(e[-2], e[-1], kernel=3, stride=2, padding=1)
        self.de5 = convt_bn_relu(e[-1], d[0], 3, 2, 1, 1)
        self.de4 = convt_bn_relu(d[0] + e[-2] if cat else d[0], d[1], 3, 2, 1, 1)
        self.de3 = convt_bn_relu(d[1] + e[-3] if cat else d[1], d[2], 3, 2, 1, 1)
        self.de2 = convt_bn_relu(d[2] + e[-4] if cat else d[2], d[3], 3, 2, 1, 1)

    def forward_encoder_layer(self, x, layer_idx=2):
        assert layer_idx > 1
        return getattr(self, f'en{layer_idx}')(x)

    def forward_decoder_layer(self, x, skip=None, skip_type='concat', layer_idx=2, cat_with=None):
        assert layer_idx > 1
        return getattr(self, f'de{layer_idx}')(x, cat_with=cat_with)


class _EqualLinearHolder(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.weight_orig = nn.Parameter(torch.randn(out_dim, in_dim))


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear = _EqualLinearHolder(in_dim, out_dim)


class AdaptiveInstanceNorm(nn.Module):
    def __init__(self, in_channel, style_dim, weighting=False):
        super().__init__()
        if weighting:
            raise NotImplementedError('adain_weighting=True is not used by the shipped configuration')
        self.norm = nn.InstanceNorm2d(in_channel)
        self.style = EqualLinear(style_dim, in_channel * 2)
        with torch.no_grad():
            self.style.linear.bias[:in_channel] = 1
            self.style.linear.bias[in_channel:] = 0
        self.weighting = False

    def forward(self, input, style):
        return Fh.wadain(input, style, self.style.linear.weight_orig, self.style.linear.bias)


class _ConvSigmoid(nn.Sequential):
    def forward(self, x):
        return run_conv(x, self[0], None, self[1], out_f32=True)


# RDFC_BRANCH_STREAMS=0: single-stream RDFGenerator (A/B: 55.9 -> 51.5 ms per train step with the second stream)
_BRANCH_STREAMS = os.environ.get('RDFC_BRANCH_STREAMS', '1') != '0'


class RDFGenerator(nn.Module):
    def __init__(self, encoder_rgb='resnet18', encoder_depth='resnet18', pretrained_on_imagenet=True,
                 semantic_channels_in=3, fuse_depth_in_rgb_decoder='WAdaIN', bn=True,
                 rgb_skip_connection_type='concat', depth_skip_connection_type='concat', adain_weighting=False,
                 rgb_channels_encoder=(64, 64, 128, 256, 512, 512), depth_channels_encoder=(64, 64, 128, 256, 512, 512),
                 rgb_channels_decoder=(256, 128, 64, 64), depth_channels_decoder=(256, 128, 64, 64),
                 use_nlspn_refine=False, nlspn_configs=None, global_guidance_module=None):
        super().__init__()
        if fuse_depth_in_rgb_decoder != 'WAdaIN' or rgb_skip_connection_type != 'concat' \
                or depth_skip_connection_type != 'concat' or not bn:
            raise NotImplementedError('HIP path covers the shipped configuration: WAdaIN fusion, concat skips, BN')
        self.use_nlspn_refine = use_nlspn_refine
        self.bn = bn
        self.rgb_skip_connection_type, self.depth_skip_connection_type = rgb_skip_connection_type, depth_skip_connection_type
        re, de = list(rgb_channels_encoder), list(depth_channels_encoder)
        rd, dd = list(rgb_channels_decoder), list(depth_channels_decoder)
        # pretrained_on_imagenet=True in the YAML only selects weights that init_weights overwrites (rdfc_gan.py:120)
        self.rgb_branch_en1 = conv_bn_relu(semantic_channels_in, re[0], kernel=3, stride=1, padding=1, bn=False)
        self.rgb_branch_encoder_decoder = EncoderDecoder(encoder_rgb, rgb_skip_connection_type, re[1:], rd, False)
        self.rgb_pred_dec1 = conv_bn_relu(64 + 64, 64, kernel=3, stride=1, padding=1)
        self.rgb_pred_dec0 = conv_bn_relu(64 + 64, 1, kernel=3, stride=1, padding=1, bn=False, relu=False)
        self.rgb_conf_dec1 = conv_bn_relu(64 + 64, 32, kernel=3, stride=1, padding=1)
        self.rgb_conf_dec0 = _ConvSigmoid(nn.Conv2d(32 + 64, 1, kernel_size=3, stride=1, padding=1), nn.Sigmoid())
        self.depth_branch_en1_rgb = conv_bn_relu(semantic_channels_in, 48, kernel=3, stride=1, padding=1, bn=False)
        self.depth_branch_en1_depth = conv_bn_relu(1, 16, kernel=3, stride=1, padding=1, bn=False)
        self.depth_branch_encoder_decoder = EncoderDecoder(encoder_depth, depth_skip_connection_type, de[1:], dd, False)
        self.id_dec1 = conv_bn_relu(64 + 64, 64, kernel=3, stride=1, padding=1)
        self.id_dec0 = conv_bn_relu(64 + 64, 1, kernel=3, stride=1, padding=1, bn=False, relu=False)
        if use_nlspn_refine:
            num_neighbors = nlspn_configs['prop_kernel'] * nlspn_configs['prop_kernel'] - 1
            self.gd_dec1 = conv_bn_relu(64 + 64, 64, kernel=3, stride=1, padding=1)
            self.gd_dec0 = conv_bn_relu(64 + 64, num_neighbors, kernel=3, stride=1, padding=1, bn=False, relu=False)
        self.cf_dec1 = conv_bn_relu(64 + 64, 32, kernel=3, stride=1, padding=1)
        self.cf_dec0 = _ConvSigmoid(nn.Conv2d(32 + 64, 1, kernel_size=3, stride=1, padding=1), nn.Sigmoid())
        if use_nlspn_refine:
            self.nlspn_refine_module = NLSPNRefineModule(**nlspn_configs)
        for i in range(1, 6):
            self.add_module(f'rgb_skip_layer{i}', nn.Identity())
        for i in range(1, 6):
            self.add_module(f'depth_skip_layer{i}', nn.Identity())
        dims = [(re[-1], re[-1]), (rd[0] + re[-2], dd[0] + de[-2]), (rd[1] + re[-3], dd[1] + re[-3]),
                (rd[2] + re[-4], dd[2] + re[-4]), (rd[3] + re[-5], dd[3] + re[-5])]
        for i, (c, cs) in enumerate(dims, 1):
            self.add_module(f'fuse_layer{i}', AdaptiveInstanceNorm(c, cs, adain_weighting))   # fuse_layer5 is unused upstream too
        self.pretrained_on_imagenet = pretrained_on_imagenet

    def forward(self, rgb, depth, normal):
        R, D = self.rgb_branch_encoder_decoder, self.depth_branch_encoder_decoder
        cat = Fh.crop_cat

        def dec_cat(net, f, lvl, skip_f):
            """decoder level + skip connection (rdf_generator.py:244-260): when the up-sampled map has the skip's size (228 = 2 x 114,
            114 = 2 x 57: the two largest levels) the level writes its channels of the concatenation itself"""
            if (2 * f.shape[1], 2 * f.shape[2]) == tuple(skip_f.shape[1:3]):
                return net.forward_decoder_layer(f, layer_idx=lvl, cat_with=skip_f)
            return cat(net.forward_decoder_layer(f, layer_idx=lvl), skip_f)
        nrm = Fh.to_pm(normal)                    # `rgb` is accepted but never read (rdf_generator.py:282)
        dep = Fh.to_pm(depth)
        # The depth branch is independent of the RGB branch (the RGB decoder only READS its features): it runs on a second
        # HIP stream so that one branch's HBM-bound normalisation kernels overlap the other's MFMA-bound convolutions.
        # Autograd replays each node on the stream of its forward, so the backward overlaps the same way.
        two = _BRANCH_STREAMS and Fh.multi_stream() and nrm.is_cuda
        main = torch.cuda.current_stream(nrm.device) if two else None
        side = Fh.branch_stream(nrm.device) if two else None
        if two:
            side.wait_stream(main)
        on_side = (lambda: torch.cuda.stream(side)) if two else contextlib.nullcontext
        # The two branches are ENQUEUED level by level in turn, not one after the other: the host launches about as fast as the
        # GPU runs the forward's short kernels, so a branch enqueued as a whole has finished on its stream before the other
        # one's first kernel arrives and nothing overlaps (kernel trace: 5.6 ms of depth branch alone, then 6 ms of RGB branch).
        # Autograd runs the nodes in reverse creation order, so the backward alternates between the streams the same way.
        # Every activation with several consumers goes through Fh.fanout: its cotangents are summed by one HIP pass.
        nh = 3 if self.use_nlspn_refine else 2          # heads on the last decoder feature / the stem feature
        with on_side():
            d1 = Fh.cat(self.depth_branch_en1_rgb(nrm), self.depth_branch_en1_depth(dep))
            d1, *d1_heads = Fh.fanout(d1, 1 + nh)
        r1, r1a, r1b = Fh.fanout(self.rgb_branch_en1(nrm), 3)
        d, skip = [None, d1], {}
        r, rskip = [None, r1], {}
        # Backward markers for the data-parallel gradient exchange (rdfc_gan.py: init_optimizer): `bwd_marks(name)` is called from inside
        # the backward when it has passed the en6 output ('dec': everything on the decoder side of this branch is final) and the input
        # of en5 ('deep': en5 + en6 are final) -- twice each, once per branch.  None: no markers (single-process runs).
        marks = self.__dict__.get('bwd_marks')

        def mark(t, name):
            if marks is None:
                return t
            m = Fh.backward_marker(t, lambda: marks(name))
            if getattr(t, '_rdfc_chain', None) is not None:
                m._rdfc_chain = t._rdfc_chain          # (the marker passes the cotangent through: the chained fan-out still applies)
            return m
        for idx in range(2, 7):
            with on_side():
                nxt = D.forward_encoder_layer(mark(d[-1], 'deep') if idx == 5 else d[-1], idx)
                if idx < 6:                                  # en2..en5 feed the next stage and a decoder skip connection
                    # (chained: the skip connection's cotangent -- a channel slice of the decoder's -- is added by the next stage's first
                    #  input-gradient launch, which already adds the block's own alias cotangent: rdfc_conv2d_dgrad_add2)
                    nxt, skip[idx] = Fh.fanout(nxt, 2, chained=True)
                else:
                    nxt = mark(nxt, 'dec')
                d.append(nxt)
            nxt = R.forward_encoder_layer(mark(r[-1], 'deep') if idx == 5 else r[-1], idx)
            if idx < 6:
                nxt, rskip[idx] = Fh.fanout(nxt, 2, chained=True)
            else:
                nxt = mark(nxt, 'dec')
            r.append(nxt)
        ready = []                                           # one event per fusion input (depth-decoder feature of a level)

        def publish(t):
            if two:
                t.record_stream(main)                        # allocated on the side stream, read on the main one
                ready.append(torch.cuda.Event())
                ready[-1].record(side)
            return t

        with on_side():
            df, ff = Fh.fanout(d[6], 2)                      # depth-decoder feature: next level | the RGB branch's fusion layer
            dfs = [publish(ff)]
        rf = r[6]
        fuses = (self.fuse_layer1, self.fuse_layer2, self.fuse_layer3, self.fuse_layer4)
        for k, lvl in enumerate((5, 4, 3, 2)):
            with on_side():                                  # depth decoder level `lvl` (its output feeds the NEXT RGB level) ...
                nxt = dec_cat(D, df, lvl, skip[lvl])
                if lvl == 2:
                    heads = Fh.fanout(nxt, nh, chained=True) # the last decoder feature only feeds the heads' first convs
                else:
                    df, ff = Fh.fanout(nxt, 2)
                    dfs.append(publish(ff))
            if two:
                main.wait_event(ready[k])                    # ... beside RGB decoder level `lvl`, which trails it by one level
            rf = dec_cat(R, fuses[k](rf, dfs[k]), lvl, rskip[lvl])
        with on_side():
            pred_init = self._head(self.id_dec0, self.id_dec1(heads[0], cat_with=d1_heads[0]), 'tanh')
            confidence = self.cf_dec0(self.cf_dec1(heads[1], cat_with=d1_heads[1]))
            if self.use_nlspn_refine:
                guide = self.gd_dec0(self.gd_dec1(heads[2], cat_with=d1_heads[2]))
                confidence, conf_prop = Fh.fanout(confidence, 2)
                depth_map_2 = self.nlspn_refine_module(pred_init, guide, conf_prop, depth)
            else:
                depth_map_2 = pred_init
        rfa, rfb = Fh.fanout(rf, 2, chained=True)
        depth_map_1 = self._head(self.rgb_pred_dec0, self.rgb_pred_dec1(rfa, cat_with=r1a), 'tanh')
        confidence_map_1 = self.rgb_conf_dec0(self.rgb_conf_dec1(rfb, cat_with=r1b))
        if two:
            for t in (depth_map_2, confidence):
                t.record_stream(main)
            main.wait_stream(side)                # depth heads (and NLSPN) done
        depth_map_2, pred = Fh.fusion_head(depth_map_1, confidence_map_1, depth_map_2, confidence)
        nchw = Fh.to_nchw
        return dict(depth_map_1=nchw(depth_map_1), confidence_map_1=nchw(confidence_map_1), depth_map_2=nchw(depth_map_2),
                    confidence_map_2=nchw(confidence), pred_depth=nchw(pred))

    @staticmethod
    def _head(seq, x, act):
        """conv(128->1)+bias followed by tanh (rdf_generator.py:373-374,383-384): fused epilogue, f32 out"""
        spec = Fh.ConvSpec(3, 1, 1, act=act, out_f32=True)
        return Fh.conv_block(x, seq[0].weight, seq[0].bias, spec=spec)


class ResnetBlock(nn.Module):
    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        if padding_type != 'reflect' or use_dropout:
            raise NotImplementedError('HIP path covers reflect padding without dropout (the shipped configuration)')
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim),
            nn.LeakyReLU(negative_slope=0.01, inplace=True),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim))

    def forward(self, x):
        cb = self.conv_block
        # ReflectionPad2d(1) + conv as ONE block: the pad is folded into the conv's loaders (functional.ConvSpec.reflect); the skip's
        # cotangent is added inside the pad's adjoint kernel in the backward
        y, xr = run_conv(x, cb[1], cb[2], cb[3], reflect=1, passthrough=True)
        return run_conv(y, cb[5], cb[6], None, res=xr, reflect=1)             # out = x + conv_block(x)


class ResnetGenerator(nn.Module):
    def __init__(self, input_channels, output_channels, ngf=64, norm_layer='BN2d', use_dropout=False, n_blocks=6,
                 padding_type='reflect'):
        super().__init__()
        assert n_blocks >= 0
        norm = nn.BatchNorm2d if norm_layer.lower() == 'bn2d' else nn.InstanceNorm2d
        use_bias = norm == nn.InstanceNorm2d
        model = [nn.ReflectionPad2d(3), nn.Conv2d(input_channels, ngf, kernel_size=7, padding=0, bias=use_bias), norm(ngf),
                 nn.LeakyReLU(negative_slope=0.01, inplace=True)]
        for i in range(2):
            mult = 2 ** i
            model += [nn.Conv2d(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=1, bias=use_bias),
                      norm(ngf * mult * 2), nn.PReLU(init=0.25)]
        for i in range(n_blocks):
            model += [ResnetBlock(ngf * 4, padding_type, norm, use_dropout, use_bias)]
        for i in range(2):
            mult = 2 ** (2 - i)
            model += [nn.ConvTranspose2d(ngf * mult, ngf * mult // 2, kernel_size=3, stride=2, padding=1, output_padding=1,
                                         bias=use_bias), norm(ngf * mult // 2), nn.PReLU(init=0.25)]
        model += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_channels, kernel_size=7, padding=0), nn.Tanh()]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        """fuses every (pad?, conv, norm?, act?) run of self.model into one HIP block"""
        h = Fh.to_pm(x)
        mods = list(self.model)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, ResnetBlock):
                h = m(h)
                i += 1
                continue
            if isinstance(m, nn.ReflectionPad2d):
                h = Fh.reflect_pad(h, m.padding[0])
                i += 1
                m = mods[i]
            assert isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)), f'unexpected layer {type(m)} at {i}'
            norm = act = None
            j = i + 1
            if j < len(mods) and isinstance(mods[j], (nn.BatchNorm2d, nn.InstanceNorm2d)):
                norm = mods[j]
                j += 1
            if j < len(mods) and isinstance(mods[j], (nn.LeakyReLU, nn.PReLU, nn.ReLU, nn.Tanh)):
                act = mods[j]
                j += 1
            last = j == len(mods)
            h = run_conv(h, m, norm, act, out_f32=last)
            i = j
        return Fh.to_nchw(h)


_GENERATORS = dict(RDFGenerator=RDFGenerator, ResnetGenerator=ResnetGenerator)


def build_generator(cfg_node):
    """cfg_node: mapping with `type` plus constructor kwargs (config/rdf_cycle_patchgan_config.yaml:3-27)."""
    kwargs = dict(cfg_node)
    typ = kwargs.pop('type')
    if typ not in _GENERATORS:
        raise KeyError(f'unknown generator type {typ}; available: {sorted(_GENERATORS)}')
    return _GENERATORS[typ](**kwargs)

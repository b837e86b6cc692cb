"""Oracle NLSPN (non-local spatial propagation).  TEST INFRASTRUCTURE ONLY.

Restates RDFC-GAN/lib/models/generator/rdf_generator/nlspn/nlspn_model.py:6-197
on top of the oracle modulated deformable conv.  Only the affinity modes with a
learned offset/affinity conv are covered ('AS', 'ASS', 'TC', 'TGASS'); the hot
path uses 'TGASS'.
"""
import torch
import torch.nn as nn

from .mdconv import MDConvOracle


class NLSPNProp(nn.Module):
    def __init__(self, channels_g, k_f=3, prop_time=18, affinity='TGASS', affinity_gamma=0.5,
                 conf_prop=True, preserve_input=False, mdconv=None):
        super().__init__()
        assert affinity in ('AS', 'ASS', 'TC', 'TGASS')
        self.k = k_f
        self.num = k_f * k_f - 1
        self.mid = self.num // 2
        self.prop_time, self.affinity = prop_time, affinity
        self.conf_prop, self.preserve_input = conf_prop, preserve_input
        self.conv_offset_aff = nn.Conv2d(channels_g, 3 * self.num, 3, 1, 1)   # nlspn_model.py:35-38
        nn.init.zeros_(self.conv_offset_aff.weight)
        nn.init.zeros_(self.conv_offset_aff.bias)
        scale = {'TC': float(self.num), 'TGASS': affinity_gamma * self.num}.get(affinity, 1.0)
        self.aff_scale_const = nn.Parameter(scale * torch.ones(1), requires_grad=(affinity == 'TGASS'))
        self.w = nn.Parameter(torch.ones(1, 1, k_f, k_f), requires_grad=False)     # :52-59
        self.b = nn.Parameter(torch.zeros(1), requires_grad=False)
        self.w_conf = nn.Parameter(torch.ones(1, 1, 1, 1), requires_grad=False)
        self.mdconv = mdconv or MDConvOracle.apply

    def offset_affinity(self, guidance, confidence):
        """nlspn_model.py:68-138."""
        B, _, H, W = guidance.shape
        n = self.num
        o1, o2, aff = torch.chunk(self.conv_offset_aff(guidance), 3, dim=1)
        # (B, n, 2, H, W) with [:, k, 0] = o1[:, k]?  No: cat(o1,o2).view(B,n,2,H,W) reinterprets
        # the 2n channels in order, so pair k = channels (2k, 2k+1) of cat(o1, o2).       :76
        off = torch.cat((o1, o2), 1).view(B, n, 2, H, W)
        zero = torch.zeros(B, 1, 2, H, W, dtype=off.dtype)
        off = torch.cat((off[:, :self.mid], zero, off[:, self.mid:]), 1).view(B, 2 * (n + 1), H, W)   # :77-80
        if self.affinity == 'TC':
            aff = torch.tanh(aff) / self.aff_scale_const
        elif self.affinity == 'TGASS':
            aff = torch.tanh(aff) / (self.aff_scale_const + 1e-8)                        # :86-87
        if self.conf_prop:                                                                # :96-118
            ones = torch.ones(B, 1, H, W, dtype=off.dtype)
            confs = []
            for t in range(n + 1):
                if t == self.mid:
                    continue
                o = off[:, 2 * t:2 * t + 2].detach()
                confs.append(self.mdconv(confidence, o, ones, self.w_conf, self.b, 1, 0, 1, 1, 1, 64))
            aff = aff * torch.cat(confs, 1)
        s = aff.abs().sum(1, keepdim=True) + 1e-4                                         # :121-122
        if self.affinity in ('ASS', 'TGASS'):
            s = torch.where(s < 1.0, torch.ones_like(s), s)                               # :124-125
        if self.affinity in ('AS', 'ASS', 'TGASS'):
            aff = aff / s
        ref = 1.0 - aff.sum(1, keepdim=True)                                              # :130-131
        aff = torch.cat((aff[:, :self.mid], ref, aff[:, self.mid:]), 1)
        return off, aff

    def forward(self, feat, guidance, confidence=None, feat_fix=None):
        off, aff = self.offset_affinity(guidance, confidence)
        if self.preserve_input:
            fixm = (feat_fix > 0).sum(1, keepdim=True).gt(0).to(feat.dtype).detach()     # :158-161
        pad = (self.k - 1) // 2
        for _ in range(self.prop_time):                                                   # :166-173
            if self.preserve_input:
                feat = (1 - fixm) * feat + fixm * feat_fix
            feat = self.mdconv(feat, off, aff, self.w, self.b, 1, pad, 1, 1, 1, 64)
        return feat, off, aff


class NLSPNRefine(nn.Module):
    """nlspn_model.py:178-197."""

    def __init__(self, prop_kernel=3, prop_time=18, affinity='TGASS', affinity_gamma=0.5,
                 conf_prop=True, preserve_input=False, mdconv=None):
        super().__init__()
        self.prop_layer = NLSPNProp(prop_kernel ** 2 - 1, prop_kernel, prop_time, affinity,
                                    affinity_gamma, conf_prop, preserve_input, mdconv)

    def forward(self, init_depth, guide, confidence, origin_depth=None):
        y, _, _ = self.prop_layer(init_depth, guide, confidence, origin_depth)
        return y

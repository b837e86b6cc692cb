"""NLSPN refinement with upstream's signatures / state_dict keys, on the HIP deformable gather.

ModulatedDeformConvFunction <- .../nlspn/modulated_deform_conv_func.py:15-56
NLPSN, NLSPNRefineModule    <- .../nlspn/nlspn_model.py:6-197

Default (prop_kernel 3, preserve_input False -- the shipped configuration): the fused kernels of csrc/nlspn.hip.
`_get_offset_affinity` is one launch behind the guidance conv (offset packing, tanh / TGASS scaling, the eight confidence
gathers, normalisation, reference weight) and the propagation loop is one autograd node: T launches forward that carry
the 1-channel map only, T launches backward for the map's cotangent, and ONE pass that produces the offset / affinity
gradients summed over all T steps (autograd over upstream's loop writes and adds 27 gradient planes per step).
Other configurations run upstream's structure step by step on rdfc_mdconv_* (ModulatedDeformConvFunction below).
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import DCN
from . import functional as Fh
from .layers import run_conv


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups, im2col_step):
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.kernel_size = _pair(weight.shape[2:4])
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        output = DCN.modulated_deform_conv_forward(input, weight, bias, offset, mask, ctx.kernel_size[0],
                                                   ctx.kernel_size[1], ctx.stride[0], ctx.stride[1], ctx.padding[0],
                                                   ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups,
                                                   ctx.deformable_groups, ctx.im2col_step)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        # upstream's extension always computes all five gradients; NLSPN freezes w / b (nlspn_model.py:55-59) and
        # detaches the offsets of the confidence gathers, so only the ones autograd asks for are launched
        need = tuple(ctx.needs_input_grad[:5])
        if all(need):
            gi, go, gm, gw, gb = DCN.modulated_deform_conv_backward(
                input, weight, bias, offset, mask, grad_output, ctx.kernel_size[0], ctx.kernel_size[1],
                ctx.stride[0], ctx.stride[1], ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups,
                ctx.deformable_groups, ctx.im2col_step)
        else:
            from . import hip
            gi, go, gm, gw, gb = hip.mdconv_backward(input.contiguous(), offset.contiguous(), mask.contiguous(), weight,
                                                     grad_output.contiguous(), ctx.stride, ctx.padding, ctx.dilation,
                                                     ctx.groups, ctx.deformable_groups, need=need)
        return gi, go, gm, gw, gb, None, None, None, None, None, None


class _Affinity(Function):
    """nlspn_model.py:68-138 behind conv_offset_aff: (offset_aff pixel-major, confidence, aff_scale_const) -> offset, aff"""

    @staticmethod
    def forward(ctx, offset_aff, confidence, scale, w_conf, b_conf, affinity):
        from . import hip
        offset, aff, cs = hip.nlspn_affinity(offset_aff, confidence, scale, w_conf, b_conf, affinity)
        ctx.save_for_backward(offset_aff, cs, scale, w_conf)
        ctx.affinity = affinity
        return offset, aff

    @staticmethod
    @once_differentiable
    def backward(ctx, g_offset, g_aff):
        from . import hip
        offset_aff, cs, scale, w_conf = ctx.saved_tensors
        if g_aff is None:
            g_aff = torch.zeros(offset_aff.shape[0], 9, offset_aff.shape[1], offset_aff.shape[2], dtype=torch.float32, device=offset_aff.device)
        g_oa, g_conf, g_scale = hip.nlspn_affinity_backward(offset_aff, cs, scale, w_conf, ctx.affinity, g_offset, g_aff,
                                                            ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return g_oa, g_conf, g_scale, None, None, None


class _Propagate(Function):
    """nlspn_model.py:166-173 (preserve_input False): T modulated-deformable steps as one autograd node"""

    @staticmethod
    def forward(ctx, feat, offset, aff, weight, bias, T):
        from . import hip
        feats = hip.nlspn_propagate(feat, offset, aff, weight, bias, T)
        ctx.save_for_backward(feats, offset, aff, weight)
        ctx.mark_non_differentiable(feats)            # the per-step maps upstream returns as `list_feat` (never used in a loss)
        return feats[T].clone(), feats

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, _g_feats):
        from . import hip
        feats, offset, aff, weight = ctx.saved_tensors
        g0, go, gm = hip.nlspn_propagate_backward(feats, offset, aff, weight, g_out.contiguous(),
                                                  ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return g0, go, gm, None, None, None


class NLPSN(nn.Module):
    def __init__(self, channels_g, channels_f, k_g, k_f, prop_time=1, affinity=None, affinity_gamma=0.5, conf_prop=True,
                 preserve_input=False):
        super().__init__()
        assert channels_f == 1, f'only tested with channels_f == 1 but {channels_f}'
        assert (k_g % 2) == 1 and (k_f % 2) == 1, 'only odd kernels are supported'
        if affinity not in ('AS', 'ASS', 'TC', 'TGASS'):
            raise NotImplementedError
        self.conf_prop, self.preserve_input = conf_prop, preserve_input
        self.prop_time, self.affinity = prop_time, affinity
        self.channels_g, self.channels_f, self.k_g, self.k_f = channels_g, channels_f, k_g, k_f
        self.num = k_f * k_f - 1
        self.idx_ref = self.num // 2
        self.conv_offset_aff = nn.Conv2d(channels_g, 3 * self.num, kernel_size=k_g, stride=1, padding=(k_g - 1) // 2, bias=True)
        self.conv_offset_aff.weight.data.zero_()
        self.conv_offset_aff.bias.data.zero_()
        if affinity == 'TC':
            self.aff_scale_const = nn.Parameter(self.num * torch.ones(1), requires_grad=False)
        elif affinity == 'TGASS':
            self.aff_scale_const = nn.Parameter(affinity_gamma * self.num * torch.ones(1))
        else:
            self.aff_scale_const = nn.Parameter(torch.ones(1), requires_grad=False)
        self.w = nn.Parameter(torch.ones((channels_f, 1, k_f, k_f)), requires_grad=False)
        self.b = nn.Parameter(torch.zeros(channels_f), requires_grad=False)
        self.w_conf = nn.Parameter(torch.ones(1, 1, 1, 1), requires_grad=False)
        self.stride, self.padding, self.dilation = 1, (k_f - 1) // 2, 1
        self.groups, self.deformable_groups, self.im2col_step = channels_f, 1, 64

    def _get_offset_affinity(self, guidance_pm, confidence=None, rgb=None):
        """guidance_pm: pixel-major (B,H,W,channels_g) in the compute dtype; confidence NCHW f32."""
        B, H, W, _ = guidance_pm.shape
        if self._fused():
            oa = run_conv(guidance_pm, self.conv_offset_aff, out_f32=True)          # pixel-major [B,H,W,24] f32
            return _Affinity.apply(oa, confidence if self.conf_prop else None, self.aff_scale_const, self.w_conf, self.b, self.affinity)
        offset_aff = Fh.to_nchw(run_conv(guidance_pm, self.conv_offset_aff, out_f32=True))
        o1, o2, aff = torch.chunk(offset_aff, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1).view(B, self.num, 2, H, W)
        zero = torch.zeros((B, 1, 2, H, W), dtype=offset.dtype, device=offset.device)
        offset = torch.cat((offset[:, :self.idx_ref], zero, offset[:, self.idx_ref:]), dim=1).view(B, -1, H, W)
        if self.affinity == 'TC':
            aff = torch.tanh(aff) / self.aff_scale_const
        elif self.affinity == 'TGASS':
            aff = torch.tanh(aff) / (self.aff_scale_const + 1e-8)
        if self.conf_prop:
            ones = torch.ones((B, 1, H, W), dtype=offset.dtype, device=offset.device)
            list_conf = []
            for idx_off in range(self.num + 1):
                if idx_off == self.idx_ref:
                    continue
                off_k = offset[:, 2 * idx_off:2 * idx_off + 2].detach().contiguous()
                list_conf.append(ModulatedDeformConvFunction.apply(confidence, off_k, ones, self.w_conf, self.b, self.stride, 0,
                                                                   self.dilation, self.groups, self.deformable_groups,
                                                                   self.im2col_step))
            aff = aff * torch.cat(list_conf, dim=1)
        aff_abs_sum = torch.sum(torch.abs(aff), dim=1, keepdim=True) + 1e-4
        if self.affinity in ('ASS', 'TGASS'):
            aff_abs_sum = torch.where(aff_abs_sum < 1.0, torch.ones_like(aff_abs_sum), aff_abs_sum)
        if self.affinity in ('AS', 'ASS', 'TGASS'):
            aff = aff / aff_abs_sum
        aff_ref = 1.0 - torch.sum(aff, dim=1, keepdim=True)
        aff = torch.cat((aff[:, :self.idx_ref], aff_ref, aff[:, self.idx_ref:]), dim=1)
        return offset.contiguous(), aff.contiguous()

    def _fused(self):
        return (self.k_f == 3 and self.channels_f == 1 and not self.preserve_input and self.conv_offset_aff.weight.is_cuda
                and os.environ.get('RDFC_NLSPN_STEPWISE') is None)

    def _propagate_once(self, feat, offset, aff):
        return ModulatedDeformConvFunction.apply(feat, offset, aff, self.w, self.b, self.stride, self.padding, self.dilation,
                                                 self.groups, self.deformable_groups, self.im2col_step)

    def forward(self, feat_init, guidance, confidence=None, feat_fix=None, rgb=None):
        assert self.channels_g == guidance.shape[3]
        assert self.channels_f == feat_init.shape[1]
        if self.conf_prop:
            assert confidence is not None
        offset, aff = self._get_offset_affinity(guidance, confidence if self.conf_prop else None, rgb)
        if self.preserve_input:
            assert feat_init.shape == feat_fix.shape
            mask_fix = (torch.sum(feat_fix > 0.0, dim=1, keepdim=True).detach() > 0.0).type_as(feat_fix)
        if self._fused():
            feat_result, feats = _Propagate.apply(feat_init.contiguous(), offset, aff, self.w, self.b, self.prop_time)
            return feat_result, list(feats[1:]), offset, aff, self.aff_scale_const.data
        feat_result = feat_init
        list_feat = []
        for _ in range(self.prop_time):
            if self.preserve_input:
                feat_result = (1.0 - mask_fix) * feat_result + mask_fix * feat_fix
            feat_result = self._propagate_once(feat_result, offset, aff)
            list_feat.append(feat_result)
        return feat_result, list_feat, offset, aff, self.aff_scale_const.data


class NLSPNRefineModule(nn.Module):
    """No encoder-decoder here.  Takes / returns 1-channel f32 maps; `guide` is pixel-major (B,H,W,8)."""

    def __init__(self, prop_kernel=3, prop_time=18, affinity='TGASS', affinity_gamma=0.5, conf_prop=True, preserve_input=False):
        super().__init__()
        self.num_neighbors = prop_kernel * prop_kernel - 1
        self.conf_prop = conf_prop
        self.prop_layer = NLPSN(self.num_neighbors, 1, 3, prop_kernel, prop_time, affinity, affinity_gamma, conf_prop,
                                preserve_input)

    def forward(self, init_pred_depth, guide, confidence, origin_depth, origin_rgb=None):
        B, H, W, _ = guide.shape
        init = init_pred_depth.reshape(B, 1, H, W)          # (B,H,W,1) and (B,1,H,W) share one layout
        conf = confidence.reshape(B, 1, H, W)
        y, _, _, _, _ = self.prop_layer(init, guide, conf, origin_depth, origin_rgb)
        return y.reshape(B, H, W, 1)

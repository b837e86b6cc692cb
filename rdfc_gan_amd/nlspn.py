"""NLSPN refinement with upstream's signatures / state_dict keys, on the HIP deformable gather.

ModulatedDeformConvFunction <- .../nlspn/modulated_deform_conv_func.py:15-56
NLPSN, NLSPNRefineModule    <- .../nlspn/nlspn_model.py:6-197

The 8 confidence gathers and the 18 propagation steps run on rdfc_mdconv_* (one
fused gather + tap-sum kernel per call, no column buffer).  The affinity
normalisation between them is still expressed with torch elementwise ops on the
(B,8,H,W) affinity tensor (fusing it into one HIP kernel is a listed next step).
"""
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
        output = DCN.modulated_deform_conv_forward(input.contiguous(), weight, bias, offset, mask, ctx.kernel_size[0],
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
                input.contiguous(), weight, bias, offset, mask, grad_output, ctx.kernel_size[0], ctx.kernel_size[1],
                ctx.stride[0], ctx.stride[1], ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups,
                ctx.deformable_groups, ctx.im2col_step)
        else:
            from . import hip
            gi, go, gm, gw, gb = hip.mdconv_backward(input.contiguous(), offset.contiguous(), mask.contiguous(), weight,
                                                     grad_output.contiguous(), ctx.stride, ctx.padding, ctx.dilation,
                                                     ctx.groups, ctx.deformable_groups, need=need)
        return gi, go, gm, gw, gb, None, None, None, None, None, None


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

"""Drop-in for upstream's pybind11 extension module `DCN`
(RDFC-GAN/lib/models/generator/rdf_generator/nlspn/deformconv/vision.cpp:5-12,
 src/modulated_deform_conv.h:10-87).

Same two entry points, argument order and error behaviour (RuntimeError on bad
arguments; CUDA/HIP tensors only -- upstream's CPU path raises "Not implement on cpu",
src/cpu/modulated_deform_cpu.cpp:25,47, and so does this one).  Put this module on
sys.path as `DCN` and upstream's modulated_deform_conv_func.py works
unchanged.  The deform_conv_* / deform_psroi_pooling_* exports of upstream are not on
the hot path (never called by the model) and are not provided.
"""
import torch

from . import hip


def _check(input, weight, bias, offset, mask):
    if not input.is_contiguous():
        raise RuntimeError('input tensor has to be contiguous')
    if not weight.is_contiguous():
        raise RuntimeError('weight tensor has to be contiguous')
    for name, t in (('input', input), ('weight', weight), ('bias', bias), ('offset', offset), ('mask', mask)):
        if not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA tensor (Not implement on cpu)')
        if t.dtype.is_floating_point and t.dtype != input.dtype:
            raise RuntimeError('all tensors must share one floating dtype')
    if input.dtype not in (torch.float32, torch.float64):      # AT_DISPATCH_FLOATING_TYPES (modulated_deform_conv_cuda.cu:94)
        raise RuntimeError(f'"modulated_deform_conv_forward_cuda" not implemented for \'{input.dtype}\'')


def modulated_deform_conv_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                                  pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, im2col_step):
    _check(input, weight, bias, offset, mask)
    if weight.shape[2] != kernel_h or weight.shape[3] != kernel_w:
        raise RuntimeError(f'Input shape and kernel shape wont match: ({weight.shape[2]} x {weight.shape[3]} vs {kernel_h} x {kernel_w}).')
    if input.shape[1] != weight.shape[1] * group:
        raise RuntimeError(f'Input shape and kernel channels wont match: ({input.shape[1]} vs {weight.shape[1] * group}).')
    step = min(input.shape[0], im2col_step)
    if input.shape[0] % step != 0:
        raise RuntimeError(f'batch({input.shape[0]}) must divide im2col_step({step})')
    return hip.mdconv_forward(input, offset.contiguous(), mask.contiguous(), weight, bias, (stride_h, stride_w),
                              (pad_h, pad_w), (dilation_h, dilation_w), group, deformable_group)


def modulated_deform_conv_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                                   pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, im2col_step):
    _check(input, weight, bias, offset, mask)
    if weight.shape[2] != kernel_h or weight.shape[3] != kernel_w:
        raise RuntimeError(f'Input shape and kernel shape wont match: ({weight.shape[2]} x {weight.shape[3]} vs {kernel_h} x {kernel_w}).')
    if input.shape[1] != weight.shape[1] * group:
        raise RuntimeError(f'Input shape and kernel channels wont match: ({input.shape[1]} vs {weight.shape[1] * group}).')
    step = min(input.shape[0], im2col_step)
    if input.shape[0] % step != 0:
        raise RuntimeError(f'batch({input.shape[0]}) must divide im2col_step({step})')
    return list(hip.mdconv_backward(input, offset.contiguous(), mask.contiguous(), weight, grad_output.contiguous(),
                                    (stride_h, stride_w), (pad_h, pad_w), (dilation_h, dilation_w), group, deformable_group))

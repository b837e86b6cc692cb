"""Oracle binding for modulated deformable convolution.  TEST INFRASTRUCTURE ONLY.

* ``forward`` / ``backward``: ctypes calls into ``mdconv_oracle.c`` (compiled with
  gcc by ``build()``), the restatement of the reference's CUDA kernels.
* ``MDConvOracle``: autograd.Function mirroring the reference's
  ``ModulatedDeformConvFunction`` (nlspn/modulated_deform_conv_func.py:15-56).
* ``forward_torch``: an independent, differentiable pure-torch formulation of the
  same sampling rule, used to cross-check the C code (values via forward,
  gradients via torch autograd).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libmdconv_oracle.so')
_lib = None


class _Shape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ('B', 'C', 'H', 'W', 'Co', 'KH', 'KW', 'sh', 'sw', 'ph', 'pw', 'dh', 'dw', 'group', 'dgroup')]


def build(force=False):
    """Compile mdconv_oracle.c -> oracle/_build/libmdconv_oracle.so (gcc -O2)."""
    src = os.path.join(_HERE, 'mdconv_oracle.c')
    inc = os.path.join(_HERE, 'mdconv_oracle_body.inc')
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(inc))):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-std=c99', '-o', _SO, src, '-lm'])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _shape(inp, weight, stride, padding, dilation, groups, dgroups):
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = inp.shape
    Co, _, KH, KW = weight.shape
    s = _Shape(B, C, H, W, Co, KH, KW, sh, sw, ph, pw, dh, dw, groups, dgroups)
    lib = _load()
    return s, lib.mdconv_oracle_out_h(ctypes.byref(s)), lib.mdconv_oracle_out_w(ctypes.byref(s))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _sfx(t):
    return {torch.float32: '_f32', torch.float64: '_f64'}[t.dtype]


def forward(inp, offset, mask, weight, bias, stride=1, padding=0, dilation=1, groups=1, dgroups=1):
    inp, offset, mask, weight = (t.detach().contiguous() for t in (inp, offset, mask, weight))
    bias = bias.detach().contiguous() if bias is not None else None
    s, Ho, Wo = _shape(inp, weight, stride, padding, dilation, groups, dgroups)
    out = torch.empty(s.B, s.Co, Ho, Wo, dtype=inp.dtype)
    getattr(_load(), 'mdconv_oracle_forward' + _sfx(inp))(
        ctypes.byref(s), _p(inp), _p(offset), _p(mask), _p(weight), _p(bias), _p(out))
    return out


def backward(inp, offset, mask, weight, grad_out, stride=1, padding=0, dilation=1, groups=1, dgroups=1):
    inp, offset, mask, weight, grad_out = (t.detach().contiguous() for t in (inp, offset, mask, weight, grad_out))
    s, _, _ = _shape(inp, weight, stride, padding, dilation, groups, dgroups)
    gi, go, gm, gw = (torch.empty_like(t) for t in (inp, offset, mask, weight))
    gb = torch.empty(s.Co, dtype=inp.dtype)
    getattr(_load(), 'mdconv_oracle_backward' + _sfx(inp))(
        ctypes.byref(s), _p(inp), _p(offset), _p(mask), _p(weight), _p(grad_out),
        _p(gi), _p(go), _p(gm), _p(gw), _p(gb))
    return gi, go, gm, gw, gb


class MDConvOracle(torch.autograd.Function):
    """Same argument order as the reference Function (modulated_deform_conv_func.py:17-18)."""

    @staticmethod
    def forward(ctx, inp, offset, mask, weight, bias, stride, padding, dilation, groups, dgroups, im2col_step):
        ctx.cfg = (stride, padding, dilation, groups, dgroups)
        ctx.save_for_backward(inp, offset, mask, weight)
        return forward(inp, offset, mask, weight, bias, *ctx.cfg)

    @staticmethod
    def backward(ctx, grad_out):
        inp, offset, mask, weight = ctx.saved_tensors
        gi, go, gm, gw, gb = backward(inp, offset, mask, weight, grad_out, *ctx.cfg)
        return gi, go, gm, gw, gb, None, None, None, None, None, None


def forward_torch(inp, offset, mask, weight, bias, stride=1, padding=0, dilation=1, groups=1, dgroups=1):
    """Independent differentiable formulation: explicit 4-corner gather with
    per-corner zeroing.  Only groups == dgroups == 1 or depthwise-free cases
    needed by the tests: general groups handled by looping."""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = inp.shape
    Co, Cg, KH, KW = weight.shape
    K = KH * KW
    Ho = (H + 2 * ph - (dh * (KH - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (KW - 1) + 1)) // sw + 1
    ys = torch.arange(Ho, dtype=inp.dtype).view(1, 1, Ho, 1) * sh - ph
    xs = torch.arange(Wo, dtype=inp.dtype).view(1, 1, 1, Wo) * sw - pw
    Cdg = C // dgroups
    cols = []   # per tap: [B, C, Ho, Wo]
    flat = inp.reshape(B, C, H * W)
    for i in range(KH):
        for j in range(KW):
            k = i * KW + j
            per_dg = []
            for dg in range(dgroups):
                oh = offset[:, dg * 2 * K + 2 * k].unsqueeze(1)
                ow = offset[:, dg * 2 * K + 2 * k + 1].unsqueeze(1)
                m = mask[:, dg * K + k].unsqueeze(1)
                hi = ys + i * dh + oh
                wi = xs + j * dw + ow
                inside = (hi > -1) & (wi > -1) & (hi < H) & (wi < W)
                h0 = torch.floor(hi)
                w0 = torch.floor(wi)
                lh, lw = hi - h0, wi - w0
                val = 0
                for dy, wy in ((0, 1 - lh), (1, lh)):
                    for dx, wx in ((0, 1 - lw), (1, lw)):
                        yy, xx = h0 + dy, w0 + dx
                        ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1) & inside
                        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long()
                        idx = idx.expand(B, Cdg, Ho, Wo).reshape(B, Cdg, Ho * Wo)
                        g = torch.gather(flat[:, dg * Cdg:(dg + 1) * Cdg], 2, idx).view(B, Cdg, Ho, Wo)
                        val = val + g * (wy * wx * ok.to(inp.dtype))
                per_dg.append(val * m)
            cols.append(torch.cat(per_dg, 1))
    col = torch.stack(cols, 2)                       # [B, C, K, Ho, Wo]
    Cog = Co // groups
    outs = []
    for g in range(groups):
        cg = col[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * K, Ho * Wo)
        wg = weight[g * Cog:(g + 1) * Cog].reshape(Cog, Cg * K)
        outs.append(torch.einsum('ok,bkp->bop', wg, cg))
    out = torch.cat(outs, 1).view(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out

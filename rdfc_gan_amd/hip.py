"""ctypes binding of include/rdfc_hip.h (librdfc_hip.so) + thin tensor-level wrappers.

Torch is used for device memory and streams only: every wrapper passes raw device
pointers, sizes and the current HIP stream to the C ABI.  There is no fallback:
if the library is missing or a kernel rejects its arguments this raises.

Feature maps are "pixel-major" 4-D torch tensors of shape [N, H, W, C] whose last
dimension is contiguous and whose pixel stride ``ld = t.stride(2)`` may exceed C
(a channel slice ``buf[..., a:b]`` of a wider map) -- see include/rdfc_hip.h.
"""
import ctypes as C
import os

import numpy as np

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RDFC_HIP_LIB', os.path.join(_HERE, 'librdfc_hip.so'))   # override: A/B builds of the kernels

F32, BF16, F16 = 0, 1, 2
# bench.py sets this to a list: every conv launch then appends (family, algorithmic flops, start, end) with
# HIP events recorded on the launch stream (per-kernel roofline measured inside the timed region)
PROFILE = None
# bench.py sets this to a list as well: the HBM-bound kernels of the normalisation layers (and NLSPN's propagation) then run as
# one C call per kernel with a HIP event pair around each: (kernel, algorithmic bytes, start, end, launches) -- the HBM-roofline
# entries of the bench line ("each reported as achieved GB/s / 8 TB/s")
PROFILE_HBM = None
ACT = dict(none=0, relu=1, lrelu=2, tanh=3, sigmoid=4, prelu=5)
_lib = None


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'KH', 'KW', 'sh', 'sw', 'ph', 'pw',
                                         'dh', 'dw', 'ldx', 'ldy', 'dtype', 'act')] + \
               [('slope', C.c_float), ('out_f32', C.c_int32)]


class MDConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'C', 'H', 'W', 'Co', 'KH', 'KW', 'sh', 'sw', 'ph', 'pw', 'dh', 'dw',
                                         'group', 'dgroup')]


def lib():
    """Load librdfc_hip.so (built ahead of time by rdfc_gan_amd.build).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: run `python -m rdfc_gan_amd.build` '
                               '(there is no CPU or PyTorch fallback for the HIP kernels)')
        _lib = C.CDLL(LIB_PATH)
        _lib.rdfc_last_error.restype = C.c_char_p
        _lib.rdfc_last_kernel.restype = C.c_char_p
        _lib.rdfc_build_hash.restype = C.c_char_p
        _declare_argtypes(_lib)
    return _lib


def _declare_argtypes(lib_):
    """argtypes for every `int rdfc_*(...)` prototype of include/rdfc_hip.h, parsed from the header that ships next to the
    library: with them ctypes converts plain Python ints / floats itself (no c_void_p / c_float objects per argument,
    ~1 800 launches x ~15 arguments per train step) and rejects a wrong argument count."""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'rdfc_hip.h')
    if not os.path.exists(hdr):
        return
    text = re.sub(r'/\*.*?\*/', ' ', open(hdr).read(), flags=re.S)
    for m in re.finditer(r'\bint\s+(rdfc_\w+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        if not hasattr(lib_, name):
            continue
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    types.append(C.c_void_p)
                elif re.match(r'^(const )?float\b', a):
                    types.append(C.c_float)
                elif re.match(r'^(const )?double\b', a):
                    types.append(C.c_double)
                elif re.match(r'^(const )?size_t\b', a):
                    types.append(C.c_size_t)
                elif re.match(r'^(const )?int64_t\b', a):
                    types.append(C.c_int64)
                elif re.match(r'^(const )?uint64_t\b', a):
                    types.append(C.c_uint64)
                else:
                    types.append(C.c_int)
        fn = getattr(lib_, name)
        fn.argtypes = types
        fn.restype = C.c_int


def build_hash():
    """hash of the kernel sources the LOADED library was compiled from (rdfc_build_hash)"""
    return lib().rdfc_build_hash().decode()


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib().rdfc_last_error().decode()}')


_dev_index = None


def _st():
    """raw handle of the current HIP stream.  torch.cuda.current_stream() costs ~9 us of Python per call (device-index and
    availability checks) and every launch needs the handle: go through the C accessor Triton uses instead."""
    global _dev_index
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_dev_index)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, 'HIP kernels need device tensors'
    return t.data_ptr()


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f'unsupported dtype {t.dtype}')


def ld(t):
    """pixel stride of a pixel-major map [N,H,W,C]"""
    if t.is_contiguous():
        return t.shape[3]
    N, H, W, Cc = t.shape
    l = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if N > 1 else Cc))
    assert t.stride(3) == 1 or Cc == 1, 'channel dim must be contiguous'
    assert (W == 1 or t.stride(2) == l) and (H == 1 or t.stride(1) == W * l) and (N == 1 or t.stride(0) == H * W * l), \
        f'not a pixel-major map: shape {tuple(t.shape)} stride {t.stride()}'
    return l


def is_pm(t):
    """whether t is a pixel-major map the kernels can read through (pointer, ld): dense, or a channel slice of a wider dense map"""
    if t.dim() != 4:
        return False
    if t.is_contiguous():
        return True
    N, H, W, Cc = t.shape
    l = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if N > 1 else Cc))
    return (t.stride(3) == 1 or Cc == 1) and l >= Cc and (W == 1 or t.stride(2) == l) and (H == 1 or t.stride(1) == W * l) \
        and (N == 1 or t.stride(0) == H * W * l)


def _f(x):
    return float(x)


# --------------------------------------------------------------------------- convolution
_desc_cache = {}


def conv_desc(x_shape, Co, k, stride, pad, dil=1, *, ldx, ldy, dtype, act='none', slope=0.0, out_f32=False,
              out_hw=None):
    """descriptor of the forward convolution; cached (a train step builds the same ~200 descriptors again and again) --
    treat the returned object as read-only"""
    key = (tuple(x_shape), Co, k, stride, pad, dil, ldx, ldy, dtype, act, slope, out_f32, out_hw)
    d = _desc_cache.get(key)
    if d is None:
        d = _desc_cache[key] = _conv_desc(x_shape, Co, k, stride, pad, dil, ldx=ldx, ldy=ldy, dtype=dtype, act=act, slope=slope,
                                          out_f32=out_f32, out_hw=out_hw)
    return d


def _conv_desc(x_shape, Co, k, stride, pad, dil=1, *, ldx, ldy, dtype, act='none', slope=0.0, out_f32=False,
               out_hw=None):
    N, Hi, Wi, Ci = x_shape
    kh, kw = (k, k) if isinstance(k, int) else k
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    dh, dw = (dil, dil) if isinstance(dil, int) else dil
    if out_hw is None:
        Ho = (Hi + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
        Wo = (Wi + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    else:
        Ho, Wo = out_hw
    return ConvDesc(N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, sh, sw, ph, pw, dh, dw, ldx, ldy, dtype, ACT[act], slope,
                    int(out_f32))


def pack_weight(w_oihw, dtype, scale=1.0, fwd=True, bwd=True):
    """-> (w_fwd [O][T][I], w_bwd [I][T][O]) in the compute dtype."""
    O, I, KH, KW = w_oihw.shape
    w = w_oihw.detach()
    assert w.dtype == torch.float32 and w.is_contiguous()
    wf = torch.empty(O, KH * KW, I, dtype=dtype, device=w.device) if fwd else None
    wb = torch.empty(I, KH * KW, O, dtype=dtype, device=w.device) if bwd else None
    _chk(lib().rdfc_pack_weight(_p(w), _p(wf), _p(wb), O, I, KH, KW, dt(wf if fwd else wb), _f(scale), _st()),
         'rdfc_pack_weight')
    return wf, wb


def is_ohwi(w):
    """4-D tensor whose memory is [O][KH][KW][I] (torch.channels_last, dense)"""
    return w.dim() == 4 and w.permute(0, 2, 3, 1).is_contiguous()


def pack_weight_ohwi(w, dtype, scale=1.0, fwd=True, bwd=True):
    """pack_weight for master weights stored channels-last (logical OIHW, memory [O][KH][KW][I])."""
    O, I, KH, KW = w.shape
    w = w.detach()
    assert w.dtype == torch.float32 and is_ohwi(w)
    wf = torch.empty(O, KH * KW, I, dtype=dtype, device=w.device) if fwd else None
    wb = torch.empty(I, KH * KW, O, dtype=dtype, device=w.device) if bwd else None
    _chk(lib().rdfc_pack_weight_ohwi(_p(w), _p(wf), _p(wb), O, I, KH, KW, dt(wf if fwd else wb), _f(scale), _st()),
         'rdfc_pack_weight_ohwi')
    return wf, wb


_PACK_JOB = np.dtype([('w', '<u8'), ('wf', '<u8'), ('wb', '<u8'), ('O', '<i4'), ('I', '<i4'), ('T', '<i4'), ('so', '<i4'), ('st', '<i4'),
                      ('si', '<i4'), ('scale', '<f4'), ('block0', '<i4')], align=True)       # rdfc_pack_job (include/rdfc_hip.h)
assert _PACK_JOB.itemsize == 56


def pack_job_table(entries, device):
    """device copy of the rdfc_pack_job array for `entries` (objects with ptr, shape, strides of the f32 weight, wf, wb, scale)
    -> (table, njobs, nblocks)"""
    tab = np.zeros(len(entries), dtype=_PACK_JOB)
    b = 0
    for k, e in enumerate(entries):
        O, I, KH, KW = e.shape
        so, si, sh, sw = e.strides
        assert KH == 1 or sh == KW * sw, 'taps must be evenly strided'
        tab[k] = (e.ptr, e.wf.data_ptr() if e.wf is not None else 0, e.wb.data_ptr() if e.wb is not None else 0, O, I, KH * KW,
                  so, sw, si, e.scale, b)
        b += KH * KW * ((O + 63) // 64) * ((I + 63) // 64)
    if not len(entries):
        return None, 0, 0
    return torch.from_numpy(tab.view(np.uint8)).to(device), len(entries), b


def pack_weights_batch(jobs, njobs, nblocks, dtype):
    _chk(lib().rdfc_pack_weights_batch(_p(jobs), njobs, nblocks, {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[dtype],
                                       _st()), 'rdfc_pack_weights_batch')


def unpack_wgrad(dw_packed, O, I, KH, KW, scale=1.0):
    g = torch.empty(O, I, KH, KW, dtype=torch.float32, device=dw_packed.device)
    _chk(lib().rdfc_unpack_wgrad(_p(dw_packed), _p(g), O, I, KH, KW, _f(scale), _st()), 'rdfc_unpack_wgrad')
    return g


def _conv_flops(d):
    return 2.0 * d.N * d.Ho * d.Wo * d.Co * d.Ci * d.KH * d.KW


class _Timed:
    """records a HIP event pair around one launch when PROFILE is active"""

    def __init__(self, family, d):
        self.on = PROFILE is not None
        if self.on:
            self.rec = [family, _conv_flops(d), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                        (d.N, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.KH, d.sh), None]

    def __enter__(self):
        if self.on:
            self.rec[2].record()

    def __exit__(self, *exc):
        if self.on:
            self.rec[3].record()
            self.rec[5] = lib().rdfc_last_kernel().decode()
            PROFILE.append(self.rec)


class _TimedHBM:
    """records (kernel, algorithmic bytes, event pair, launches) around one call when PROFILE_HBM is active"""

    def __init__(self, kernel, nbytes, launches=1):
        self.on = PROFILE_HBM is not None
        if self.on:
            self.rec = [kernel, float(nbytes), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), launches]

    def __enter__(self):
        if self.on:
            self.rec[2].record()

    def __exit__(self, *exc):
        if self.on:
            self.rec[3].record()
            PROFILE_HBM.append(self.rec)


def conv_forward(d, x, w_fwd, bias, y, reflect=False):
    """reflect: d.ph = d.pw is a REFLECTION padding (conv_reflect_supported(d) must hold)"""
    with _Timed('igemm_kernel (conv fwd / dgrad)', d):
        if reflect:
            _chk(lib().rdfc_conv2d_forward_reflect(C.byref(d), _p(x), _p(w_fwd), _p(bias), _p(y), _st()), 'rdfc_conv2d_forward_reflect')
        else:
            _chk(lib().rdfc_conv2d_forward(C.byref(d), _p(x), _p(w_fwd), _p(bias), _p(y), _st()), 'rdfc_conv2d_forward')
    return y


_reflect_ok = {}


def conv_reflect_supported(d):
    """whether forward and weight gradient of d run with the reflection padding folded into their loaders (cached per geometry)"""
    key = (d.N, d.Hi, d.Wi, d.Ci, d.Co, d.KH, d.sh, d.ph, d.ldx, d.ldy, d.dtype, d.act, d.out_f32)
    hit = _reflect_ok.get(key)
    if hit is None:
        hit = _reflect_ok[key] = bool(lib().rdfc_conv2d_reflect_supported(C.byref(d)))
    return hit


def conv_stats_nchunk(d, G):
    return lib().rdfc_conv_stats_nchunk(C.byref(d), int(G))


def conv_forward_partials(d, x, w_fwd, bias, y, G):
    """conv forward with the normalisation statistics fused into its epilogue (n = conv_stats_nchunk(d, G) must be > 0): G groups of
    N / G consecutive images.  -> (buf, n): buf = f32 [partial rows G * n * 2 * Co | f64 scratch [G][2][Co] for norm_from_partials]"""
    n = conv_stats_nchunk(d, G)
    assert n > 0
    npart = (G * n * 2 * d.Co + 1) & ~1
    buf = torch.empty(npart + 4 * G * d.Co, dtype=torch.float32, device=x.device)
    with _Timed('igemm_kernel (conv fwd / dgrad)', d):
        _chk(lib().rdfc_conv2d_forward_stats(C.byref(d), _p(x), _p(w_fwd), _p(bias), _p(y), _p(buf), int(G), _st()),
             'rdfc_conv2d_forward_stats')
    return buf, n


def norm_from_partials(buf, n, y, G, eps, gamma, beta, res, z, act='none', slope=0.0, prelu=None, run_mean=None, run_var=None,
                       momenta=None):
    """statistics per group from the partial rows of conv_forward_partials (one small workgroup per (group, 32 channels)), then the
    normalise (+ residual + activation) pass, which also advances the running statistics once per group in group order (momenta: G
    floats; BatchNorm only).  -> mean[G,C], rstd[G,C]; z is written."""
    N, H, W, Cc = y.shape
    P = N * H * W // G
    npart = (G * n * 2 * Cc + 1) & ~1
    stats = torch.empty(2, G, Cc, dtype=torch.float32, device=y.device)
    base = buf.data_ptr()
    _chk(lib().rdfc_norm_finalize_apply_seq(base, int(n), base + 4 * npart, _p(y), _p(stats[0]), _p(stats[1]), _p(gamma), _p(beta), _p(res),
                                            _p(z), G, P, Cc, ld(y), ld(res) if res is not None else 0, ld(z), dt(y), _f(eps), _p(run_mean),
                                            _p(run_var), _momenta(momenta) if momenta is not None else None, ACT[act], _f(slope), _p(prelu),
                                            _st()), 'rdfc_norm_finalize_apply_seq')
    return stats[0], stats[1]


def conv_forward_stats(d, x, w_fwd, bias, y, G, eps, run_mean=None, run_var=None, momentum=0.1, group_momenta=None):
    """conv forward with fused statistics, finalized: -> mean[G,Co], rstd[G,Co] (G = 1 with group_momenta of length Gs: BatchNorm over
    Gs stacked batches, -> [Gs,Co]).  (The train step uses conv_forward_partials + norm_from_partials; this form serves the tests.)"""
    Gs = len(group_momenta) if group_momenta is not None else G
    buf, n = conv_forward_partials(d, x, w_fwd, bias, y, Gs)
    mean = torch.empty(Gs, d.Co, dtype=torch.float32, device=x.device)
    rstd = torch.empty(Gs, d.Co, dtype=torch.float32, device=x.device)
    P = d.N * d.Ho * d.Wo // Gs
    if group_momenta is not None:
        _chk(lib().rdfc_norm_finalize_seq(_p(buf), n, _p(mean), _p(rstd), Gs, P, d.Co, dt(x), _f(eps),
                                          _p(run_mean), _p(run_var), _momenta(group_momenta), _st()), 'rdfc_norm_finalize_seq')
    else:
        _chk(lib().rdfc_norm_finalize_n(_p(buf), n, _p(mean), _p(rstd), int(G), P, d.Co, _f(eps), _p(run_mean), _p(run_var),
                                        _f(momentum), _st()), 'rdfc_norm_finalize_n')
    return mean, rstd


def conv_dgrad(d, dy, w_bwd, bias, dx, addend=None, addend2=None):
    """addend (, addend2): maps like dx (may be channel slices) added to the input gradient in the same pass (rdfc_conv2d_dgrad_add / _add2)"""
    if addend is None and addend2 is not None:
        addend, addend2 = addend2, None
    with _Timed('igemm_kernel (conv fwd / dgrad)', d):
        if addend is None:
            _chk(lib().rdfc_conv2d_dgrad(C.byref(d), _p(dy), _p(w_bwd), _p(bias), _p(dx), _st()), 'rdfc_conv2d_dgrad')
        elif addend2 is not None:
            assert bias is None and addend.dtype == dx.dtype == addend2.dtype and addend.shape == dx.shape == addend2.shape
            _chk(lib().rdfc_conv2d_dgrad_add2(C.byref(d), _p(dy), _p(w_bwd), _p(addend), ld(addend), _p(addend2), ld(addend2), _p(dx), _st()),
                 'rdfc_conv2d_dgrad_add2')
        else:
            assert bias is None and addend.dtype == dx.dtype and addend.shape == dx.shape
            _chk(lib().rdfc_conv2d_dgrad_add(C.byref(d), _p(dy), _p(w_bwd), _p(addend), ld(addend), _p(dx), _st()), 'rdfc_conv2d_dgrad_add')
    return dx


def conv_wgrad(d, x, dy, into=None, stream=None, reflect=False):
    """into: f32 buffer [Co][KH*KW][Ci] to ACCUMULATE into (e.g. a channels-last .grad); default a fresh zeroed one.
    stream: raw HIP stream handle to launch on (default: the current stream)
    reflect: d.ph = d.pw is a REFLECTION padding of x (conv_reflect_supported(d) must hold)"""
    dw = into if into is not None else torch.zeros(d.Co, d.KH * d.KW, d.Ci, dtype=torch.float32, device=x.device)
    fn, name = (lib().rdfc_conv2d_wgrad_reflect, 'rdfc_conv2d_wgrad_reflect') if reflect else (lib().rdfc_conv2d_wgrad, 'rdfc_conv2d_wgrad')
    if stream is not None and PROFILE is None:
        _chk(fn(C.byref(d), _p(x), _p(dy), _p(dw), stream), name)
        return dw
    with _Timed('wgrad_kernel (conv weight grad)', d):
        _chk(fn(C.byref(d), _p(x), _p(dy), _p(dw), _st() if stream is None else stream), name)
    return dw


def conv_wgrad_act_supported(d):
    """whether conv_wgrad_act covers the layer (thin input, stride 1, ReLU / LeakyReLU in d.act)"""
    return bool(lib().rdfc_conv2d_wgrad_act_supported(C.byref(d)))


def conv_wgrad_act(d, x, dz, y_act, into, bias_into=None, stream=None):
    """weight (and bias) gradient of a conv block without normalisation, its activation's backward folded in: dz is the cotangent behind
    the activation named by d.act / d.slope, y_act the block's activated output; accumulates into `into` [Co][KH*KW][Ci] (and bias_into)"""
    with _Timed('wgrad_kernel (conv weight grad)', d):
        _chk(lib().rdfc_conv2d_wgrad_act(C.byref(d), _p(x), _p(dz), _p(y_act), ld(y_act), _p(into), _p(bias_into),
                                         _st() if stream is None else stream), 'rdfc_conv2d_wgrad_act')
    return into


def wgrad_ring_reserve(stream, nbytes=0):
    """give `stream` (a torch.cuda.Stream) the deterministic weight-gradient kernels' scratch slab before it is captured into a graph"""
    _chk(lib().rdfc_wgrad_ring_reserve(stream.cuda_stream, C.c_size_t(nbytes)), 'rdfc_wgrad_ring_reserve')


def wgrad_is_deterministic(d, reflect=False):
    """whether conv_wgrad(d, ...) sums its split-K partials in a fixed order (bitwise reproducible) or with float atomics"""
    return bool(lib().rdfc_conv2d_wgrad_deterministic(C.byref(d), int(reflect)))


def zero_(t):
    """hipMemsetAsync on the current stream (a contiguous tensor or a contiguous slice of one)"""
    assert t.is_contiguous()
    _chk(lib().rdfc_zero(_p(t), C.c_size_t(t.numel() * t.element_size()), _st()), 'rdfc_zero')
    return t


# --------------------------------------------------------------------------- layout
def nchw_to_pm(src, dtype, out=None):
    N, Cc, H, W = src.shape
    src = src.contiguous()
    assert src.dtype == torch.float32
    if out is None:
        out = torch.empty(N, H, W, Cc, dtype=dtype, device=src.device)
    _chk(lib().rdfc_nchw_to_pm(_p(src), _p(out), N, Cc, H, W, ld(out), dt(out), _st()), 'rdfc_nchw_to_pm')
    return out


def pm_to_nchw(src):
    N, H, W, Cc = src.shape
    out = torch.empty(N, Cc, H, W, dtype=torch.float32, device=src.device)
    _chk(lib().rdfc_pm_to_nchw(_p(src), _p(out), N, Cc, H, W, ld(src), dt(src), _st()), 'rdfc_pm_to_nchw')
    return out


def copy_window(src, dst, accumulate=False):
    """copy the common top-left window of two maps (crop when dst is smaller, zero-extension's
    data part when dst is larger)"""
    N, Hs, Ws, Cc = src.shape
    _, Hd, Wd, Cd = dst.shape
    assert Cc == Cd and src.dtype == dst.dtype
    _chk(lib().rdfc_copy_window(_p(src), _p(dst), N, Hs, Ws, Hd, Wd, min(Hs, Hd), min(Ws, Wd), Cc, ld(src), ld(dst),
                                dt(src), int(accumulate), _st()), 'rdfc_copy_window')
    return dst


def reflect_pad(src, pad):
    N, H, W, Cc = src.shape
    assert src.is_contiguous()
    out = torch.empty(N, H + 2 * pad, W + 2 * pad, Cc, dtype=src.dtype, device=src.device)
    _chk(lib().rdfc_reflect_pad(_p(src), _p(out), N, H, W, Cc, pad, dt(src), _st()), 'rdfc_reflect_pad')
    return out


def reflect_pad_adjoint(dpad, pad, addend=None):
    N, Hp, Wp, Cc = dpad.shape
    assert dpad.is_contiguous()
    H, W = Hp - 2 * pad, Wp - 2 * pad
    out = torch.empty(N, H, W, Cc, dtype=dpad.dtype, device=dpad.device)
    if addend is None:
        _chk(lib().rdfc_reflect_pad_adjoint(_p(dpad), _p(out), N, H, W, Cc, pad, dt(dpad), _st()), 'rdfc_reflect_pad_adjoint')
    else:
        assert addend.dtype == dpad.dtype and tuple(addend.shape) == (N, H, W, Cc)
        _chk(lib().rdfc_reflect_pad_adjoint_add(_p(dpad), _p(addend), ld(addend), _p(out), N, H, W, Cc, pad, dt(dpad), _st()),
             'rdfc_reflect_pad_adjoint_add')
    return out


# --------------------------------------------------------------------------- normalisation
_nchunk_cache = {}


def norm_nchunk(P, Cc, dtype_code):
    key = (P, Cc, dtype_code)
    n = _nchunk_cache.get(key)
    if n is None:
        n = _nchunk_cache[key] = lib().rdfc_norm_nchunk(P, Cc, dtype_code)
    return n


_scratch_cache = {}


def norm_scratch(G, nchunk, K, Cc):
    """f32 elements of scratch a two-stage reduction needs (partial rows + the f64 per-group statistics of a stacked-batch BatchNorm)"""
    key = (G, nchunk, K, Cc)
    n = _scratch_cache.get(key)
    if n is None:
        n = _scratch_cache[key] = lib().rdfc_norm_scratch_floats(G, nchunk, K, Cc)
    return n


_mom_arrays = {}


def _momenta(group_momenta):
    """host float[G] for rdfc_norm_*_seq (cached per value tuple)"""
    key = tuple(float(m) for m in group_momenta)
    a = _mom_arrays.get(key)
    if a is None:
        a = _mom_arrays[key] = (C.c_float * len(key))(*key)
    return a


def norm_forward(x, G, eps, gamma, beta, res, z, act='none', slope=0.0, prelu=None, run_mean=None, run_var=None, momentum=0.1,
                 group_momenta=None):
    """statistics + finalize + normalise(+residual)(+activation) in one call -> (mean[G,C], rstd[G,C]); z is written.
    group_momenta (G floats): BatchNorm over G stacked batches -- the running statistics advance once per group, in order"""
    N, H, W, Cc = x.shape
    P = N * H * W // G
    d = dt(x)
    npart = norm_scratch(G, norm_nchunk(P, Cc, d), 2, Cc)
    buf = torch.empty(npart + 2 * G * Cc, dtype=torch.float32, device=x.device)      # [partials, level-2 rows | mean | rstd]
    base = buf.data_ptr()
    if PROFILE_HBM is not None:
        # one call and one event pair per kernel (same kernels and results as the composites below, which stay un-timed)
        mb = x.numel() * x.element_size()
        pm, pr = base + 4 * npart, base + 4 * (npart + G * Cc)
        with _TimedHBM('stats_kernel', mb):
            _chk(lib().rdfc_norm_stats(_p(x), base, G, P, Cc, ld(x), d, _st()), 'rdfc_norm_stats')
        if group_momenta is not None:
            _chk(lib().rdfc_norm_finalize_seq(base, 0, pm, pr, G, P, Cc, d, _f(eps), _p(run_mean), _p(run_var), _momenta(group_momenta),
                                              _st()), 'rdfc_norm_finalize_seq')
        else:
            _chk(lib().rdfc_norm_finalize(base, pm, pr, G, P, Cc, d, _f(eps), _p(run_mean), _p(run_var), _f(momentum), _st()),
                 'rdfc_norm_finalize')
        with _TimedHBM('apply_kernel', mb * (2 + (res is not None))):
            _chk(lib().rdfc_norm_apply(_p(x), pm, pr, _p(gamma), _p(beta), _p(res), _p(z), G, P, Cc, ld(x),
                                       ld(res) if res is not None else 0, ld(z), d, ACT[act], _f(slope), _p(prelu), _st()), 'rdfc_norm_apply')
        stats = buf[npart:].view(2, G, Cc)
        return stats[0], stats[1]
    if group_momenta is not None:
        assert len(group_momenta) == G
        _chk(lib().rdfc_norm_forward_seq(_p(x), base, base + 4 * npart, base + 4 * (npart + G * Cc), _p(gamma), _p(beta), _p(res), _p(z),
                                         G, P, Cc, ld(x), ld(res) if res is not None else 0, ld(z), d, _f(eps), _p(run_mean),
                                         _p(run_var), _momenta(group_momenta), ACT[act], _f(slope), _p(prelu), _st()),
             'rdfc_norm_forward_seq')
        stats = buf[npart:].view(2, G, Cc)
        return stats[0], stats[1]
    _chk(lib().rdfc_norm_forward(_p(x), base, base + 4 * npart, base + 4 * (npart + G * Cc), _p(gamma), _p(beta), _p(res), _p(z), G, P,
                                 Cc, ld(x), ld(res) if res is not None else 0, ld(z), d, _f(eps), _p(run_mean), _p(run_var),
                                 _f(momentum), ACT[act], _f(slope), _p(prelu), _st()), 'rdfc_norm_forward')
    stats = buf[npart:].view(2, G, Cc)
    return stats[0], stats[1]


def norm_stats(x, G, eps, run_mean=None, run_var=None, momentum=0.1):
    """x [N,H,W,C]; G = 1 (batch) or N (instance). -> mean[G,C], rstd[G,C]"""
    N, H, W, Cc = x.shape
    P = N * H * W // G
    npart = norm_scratch(G, norm_nchunk(P, Cc, dt(x)), 2, Cc)
    buf = torch.empty(npart + 2 * G * Cc, dtype=torch.float32, device=x.device)      # [partials, level-2 rows | mean | rstd]
    base = buf.data_ptr()
    _chk(lib().rdfc_norm_stats_final(_p(x), base, base + 4 * npart, base + 4 * (npart + G * Cc), G, P, Cc, ld(x), dt(x), _f(eps),
                                     _p(run_mean), _p(run_var), _f(momentum), _st()), 'rdfc_norm_stats_final')
    stats = buf[npart:].view(2, G, Cc)
    return stats[0], stats[1]


def norm_apply(x, mean, rstd, gamma, beta, res, z, act='none', slope=0.0, prelu=None):
    N, H, W, Cc = x.shape
    G = mean.shape[0]
    P = N * H * W // G
    _chk(lib().rdfc_norm_apply(_p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), _p(z), G, P, Cc, ld(x),
                               ld(res) if res is not None else 0, ld(z), dt(x), ACT[act], _f(slope), _p(prelu), _st()),
         'rdfc_norm_apply')
    return z


def norm_backward(dz, x, mean, rstd, gamma, beta, res, act='none', slope=0.0, prelu=None, frozen=False,
                  want_dres=False, acc_beta=None, acc_gamma=None, acc_bias=None):
    """-> dx, dres (or None), sums[G,3,C] (sum du, sum du*xhat, sum dz*min(u,0)).
    acc_beta / acc_gamma: f32[C] buffers that additionally receive += the sums over all groups of sums[g,0] / sums[g,1].
    acc_bias: f32[C] buffer that receives += the sum over pixels of dx as stored (the gradient of a conv bias in front of the norm
    layer: round 6, rdfc_norm_backward_dxsum -- one pass over dx less than channel_sum(dx, acc=...)).
    One call: reduce + finalize + apply (rdfc_norm_backward)."""
    N, H, W, Cc = x.shape
    G = mean.shape[0]
    P = N * H * W // G
    d = dt(x)
    npart = norm_scratch(G, norm_nchunk(P, Cc, d), 3, Cc)
    buf = torch.empty(npart + G * 3 * Cc, dtype=torch.float32, device=x.device)       # [partials, level-2 rows | sums]
    base = buf.data_ptr()
    dx = torch.empty(N, H, W, Cc, dtype=x.dtype, device=x.device)
    dres = torch.empty(N, H, W, Cc, dtype=x.dtype, device=x.device) if want_dres else None
    dxs = None
    if acc_bias is not None:
        dxs = torch.empty(lib().rdfc_norm_dxsum_scratch_floats(G, P, Cc, d), dtype=torch.float32, device=x.device)
    if PROFILE_HBM is not None:
        mb, nr = x.numel() * x.element_size(), 2 + (res is not None)
        ps = base + 4 * npart
        with _TimedHBM('bwd_reduce_kernel', mb * nr):
            _chk(lib().rdfc_norm_bwd_reduce(_p(dz), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), base, G, P, Cc, ld(dz), ld(x),
                                            ld(res) if res is not None else 0, d, ACT[act], _f(slope), _p(prelu), _st()), 'rdfc_norm_bwd_reduce')
        _chk(lib().rdfc_norm_bwd_finalize(base, ps, G, P, Cc, d, _st()), 'rdfc_norm_bwd_finalize')
        with _TimedHBM('bwd_apply_kernel', mb * (nr + 1 + bool(want_dres))):
            _chk(lib().rdfc_norm_bwd_apply_dxsum(_p(dz), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), ps, _p(dx), _p(dres), G, P, Cc,
                                                 ld(dz), ld(x), ld(res) if res is not None else 0, Cc, Cc, d, ACT[act], _f(slope), _p(prelu),
                                                 int(frozen), _p(acc_beta), _p(acc_gamma), _p(dxs), _p(acc_bias), _st()), 'rdfc_norm_bwd_apply_dxsum')
        return dx, dres, buf[npart:].view(G, 3, Cc)
    if acc_bias is not None:
        _chk(lib().rdfc_norm_backward_dxsum(_p(dz), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), base, base + 4 * npart, _p(dx),
                                            _p(dres), G, P, Cc, ld(dz), ld(x), ld(res) if res is not None else 0, Cc, Cc, d, ACT[act], _f(slope),
                                            _p(prelu), int(frozen), _p(acc_beta), _p(acc_gamma), _p(dxs), _p(acc_bias), _st()), 'rdfc_norm_backward_dxsum')
        return dx, dres, buf[npart:].view(G, 3, Cc)
    _chk(lib().rdfc_norm_backward(_p(dz), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(res), base, base + 4 * npart, _p(dx),
                                  _p(dres), G, P, Cc, ld(dz), ld(x), ld(res) if res is not None else 0, Cc, Cc, d, ACT[act], _f(slope),
                                  _p(prelu), int(frozen), _p(acc_beta), _p(acc_gamma), _st()), 'rdfc_norm_backward')
    return dx, dres, buf[npart:].view(G, 3, Cc)


def adain_apply(x, mean, rstd, gb, out):
    N, H, W, Cc = x.shape
    _chk(lib().rdfc_adain_apply(_p(x), _p(mean), _p(rstd), _p(gb), _p(out), N, H * W, Cc, ld(x), ld(gb), ld(out), dt(x), _st()),
         'rdfc_adain_apply')
    return out


def adain_backward(dout, x, mean, rstd, gb):
    N, H, W, Cc = x.shape
    P = H * W
    npart = norm_scratch(N, norm_nchunk(P, Cc, dt(x)), 3, Cc)
    buf = torch.empty(npart + N * 3 * Cc, dtype=torch.float32, device=x.device)       # [partials, level-2 rows | sums]
    base = buf.data_ptr()
    dx = torch.empty(N, H, W, Cc, dtype=x.dtype, device=x.device)
    dgb = torch.empty(N, H, W, 2 * Cc, dtype=x.dtype, device=x.device)
    _chk(lib().rdfc_adain_backward(_p(dout), _p(x), _p(mean), _p(rstd), _p(gb), base, base + 4 * npart, _p(dx), _p(dgb), N, P, Cc,
                                   ld(dout), ld(x), ld(gb), Cc, 2 * Cc, dt(x), _st()), 'rdfc_adain_backward')
    return dx, dgb


def channel_sum(x, acc=None):
    """out[c] = sum over pixels; acc (f32 [C], optional): acc += out in the same launch"""
    N, H, W, Cc = x.shape
    P = N * H * W
    partial = torch.empty(lib().rdfc_channel_sum_scratch_floats(P, Cc, ld(x), dt(x)), dtype=torch.float32, device=x.device)
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    _chk(lib().rdfc_channel_sum_acc(_p(x), _p(partial), _p(out), _p(acc), P, Cc, ld(x), dt(x), _st()), 'rdfc_channel_sum_acc')
    return out


def act_bwd(dy, y, act, slope=0.0):
    N, H, W, Cc = dy.shape
    dx = torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    _chk(lib().rdfc_act_bwd(_p(dy), _p(y), _p(dx), N * H * W, Cc, ld(dy), ld(y), Cc, dt(dy), int(y.dtype == torch.float32),
                            ACT[act], _f(slope), _st()), 'rdfc_act_bwd')
    return dx


# --------------------------------------------------------------------------- heads, losses, optimizer
def fusion_head(d1, c1, d2, c2):
    d2c, pred = torch.empty_like(d2), torch.empty_like(d2)
    _chk(lib().rdfc_fusion_head(_p(d1), _p(c1), _p(d2), _p(c2), _p(d2c), _p(pred), d1.numel(), _st()), 'rdfc_fusion_head')
    return d2c, pred


def fusion_head_bwd(d1, c1, d2, c2, g_pred, g_d2c):
    outs = [torch.empty_like(d1) for _ in range(4)]
    _chk(lib().rdfc_fusion_head_bwd(_p(d1), _p(c1), _p(d2), _p(c2), _p(g_pred), _p(g_d2c), *[_p(o) for o in outs],
                                    d1.numel(), _st()), 'rdfc_fusion_head_bwd')
    return outs


def weighted_loss(pred, target, weight, kind, want_grad=True, grad_scale=1.0):
    """pred [N,C,H,W] f32 (upstream layout); target tensor or python float; weight [N,1,H,W] / [N,H,W] or None."""
    N, Cc, H, W = pred.shape
    pred = pred.contiguous()
    numel = pred.numel()
    tconst, tt = (float(target), None) if not torch.is_tensor(target) else (0.0, target.contiguous())
    uw = 0.0
    if weight is None:
        uw = 1.0 / (numel + 1e-6)
    else:
        weight = weight.to(torch.float32).contiguous()
        assert weight.numel() == N * H * W, 'weight must broadcast over channels'
    nch = lib().rdfc_loss_nchunk(numel)
    partial = torch.empty(nch, dtype=torch.float32, device=pred.device)
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    _chk(lib().rdfc_weighted_loss(_p(pred), _p(tt), _f(tconst), _p(weight), _f(uw), N, Cc, H * W, kind, _p(partial), _p(out),
                                  _p(grad), _f(grad_scale), _st()), 'rdfc_weighted_loss')
    return out, grad


def adam_step(param, grad, m, v, lr, b1, b2, eps, step, grad_scale=1.0):
    _chk(lib().rdfc_adam_step(_p(param), _p(grad), _p(m), _p(v), C.c_size_t(param.numel()), _f(lr), _f(b1), _f(b2), _f(eps),
                              int(step), _f(grad_scale), _st()), 'rdfc_adam_step')


def adamw_step(param, grad, m, v, lr, b1, b2, eps, weight_decay, step, grad_scale=1.0):
    _chk(lib().rdfc_adamw_step(_p(param), _p(grad), _p(m), _p(v), C.c_size_t(param.numel()), _f(lr), _f(b1), _f(b2), _f(eps),
                               _f(weight_decay), int(step), _f(grad_scale), _st()), 'rdfc_adamw_step')


def adam_step_dev(param, grad, m, v, hyper, b1, b2, eps, grad_scale=1.0):
    """hyper: device f32 tensor {lr, step}"""
    _chk(lib().rdfc_adam_step_dev(_p(param), _p(grad), _p(m), _p(v), C.c_size_t(param.numel()), _p(hyper), _f(b1), _f(b2),
                                  _f(eps), _f(grad_scale), _st()), 'rdfc_adam_step_dev')


def grad_nonfinite(grad, ostate):
    """ostate[0] <- 1 when the flat gradient holds an inf / NaN, else 0 (device side, no read-back)"""
    _chk(lib().rdfc_grad_nonfinite(_p(grad), C.c_size_t(grad.numel()), _p(ostate), _st()), 'rdfc_grad_nonfinite')


def adam_step_scaled(param, grad, m, v, lr, b1, b2, eps, step, scaler, ostate, grad_scale=1.0, weight_decay=0.0, hyper=None):
    """Adam (weight_decay 0) / AdamW under dynamic loss scaling: include/rdfc_hip.h, rdfc_adam_step_scaled"""
    _chk(lib().rdfc_adam_step_scaled(_p(param), _p(grad), _p(m), _p(v), C.c_size_t(param.numel()), _f(lr), _p(hyper), int(step), _f(b1),
                                     _f(b2), _f(eps), _f(weight_decay), _f(grad_scale), _p(scaler), _p(ostate), _st()), 'rdfc_adam_step_scaled')


def loss_scale_update(scaler, growth, backoff, interval, min_scale, max_scale):
    _chk(lib().rdfc_loss_scale_update(_p(scaler), _f(growth), _f(backoff), int(interval), _f(min_scale), _f(max_scale), _st()),
         'rdfc_loss_scale_update')


def depth_metrics(gt_norm, pred_norm, evaluate_mask=None, std=5.0, mean=5.0, t_valid=1e-4):
    """normalised (N,1,H,W) | (N,H,W) f32 maps -> per-image metrics [N,8] (RMSE, MAE, iRMSE, iMAE, REL, d1, d2, d3)"""
    g, p_ = gt_norm.contiguous().float(), pred_norm.contiguous().float()
    N = g.shape[0]
    HW = g.numel() // N
    assert p_.numel() == g.numel()
    m = None
    if evaluate_mask is not None:
        m = evaluate_mask.contiguous().to(torch.uint8)
        assert m.numel() == g.numel()
    out = torch.empty(N, 8, dtype=torch.float32, device=g.device)
    _chk(lib().rdfc_depth_metrics(_p(g), _p(p_), _p(m), N, HW, _f(std), _f(mean), _f(t_valid), _p(out), _st()), 'rdfc_depth_metrics')
    return out


def draw_pepper_mask(N, H, W, seed, counter, snr, device):
    mask = torch.empty(N, 1, H, W, dtype=torch.uint8, device=device)
    _chk(lib().rdfc_draw_pepper_mask(_p(mask), N, H * W, int(seed) & (2 ** 64 - 1), int(counter), _f(snr), _st()), 'rdfc_draw_pepper_mask')
    return mask


def mask_keep(g, mask_u8):
    """g [N,C,H,W] f32 * (mask == 0): the backward of apply_pepper"""
    N, Cc, H, W = g.shape
    g = g.contiguous()
    out = torch.empty_like(g)
    _chk(lib().rdfc_mask_keep(_p(g), _p(mask_u8), _p(out), N, Cc, H * W, _st()), 'rdfc_mask_keep')
    return out


def scale_dev(x, scalar, k=1.0):
    """x (f32, contiguous) * scalar (0-dim device f32 tensor or None) * k"""
    assert x.dtype == torch.float32 and x.is_contiguous()
    if scalar is not None:
        scalar = scalar.reshape(()).to(torch.float32)
    out = torch.empty_like(x)
    _chk(lib().rdfc_scale_dev(_p(x), _p(scalar), _f(k), _p(out), C.c_size_t(x.numel()), _st()), 'rdfc_scale_dev')
    return out


def unit_normals(v):
    """helper.py:404-408 on channels 0..2 of an NCHW f32 map (no gradient)"""
    v = v.detach().contiguous().float()
    N, Cs, H, W = v.shape
    out = torch.empty(N, 3, H, W, dtype=torch.float32, device=v.device)
    _chk(lib().rdfc_unit_normals(_p(v), _p(out), N, Cs, H * W, _st()), 'rdfc_unit_normals')
    return out


def sum_flat(tensors):
    """sum of 2..4 contiguous tensors of one shape / dtype (f32 | compute dtype) in one pass"""
    t0 = tensors[0]
    n = t0.numel()
    vec = 8 if t0.dtype != torch.float32 else 4
    Cc = vec if n % vec == 0 else 1
    views = [t.contiguous().view(1, 1, n // Cc, Cc) for t in tensors]
    return sum_maps(views).view(t0.shape)


def sum_maps(maps, out=None):
    """sum of 1..4 pixel-major maps [N,H,W,C] (each may be a channel slice of a wider map) in one pass"""
    m0 = maps[0]
    N, H, W, Cc = m0.shape
    if out is None:
        out = torch.empty(N, H, W, Cc, dtype=m0.dtype, device=m0.device)
    n = len(maps)
    assert 1 <= n <= 4 and all(m.shape == m0.shape and m.dtype == m0.dtype for m in maps)
    ptrs = (C.c_void_p * n)(*[m.data_ptr() for m in maps])
    lds = (C.c_int * n)(*[ld(m) for m in maps])
    _chk(lib().rdfc_sum_maps(n, ptrs, lds, _p(out), N * H * W, Cc, ld(out), dt(m0), _st()), 'rdfc_sum_maps')
    return out


def combine_scalars(terms, weights, post=1.0):
    """terms: 0-dim f32 device tensors -> f32[n + 1] = (w_i * l_i ..., post * sum)"""
    n = len(terms)
    ts = [t.detach().reshape(()).float() for t in terms]
    out = torch.empty(n + 1, dtype=torch.float32, device=ts[0].device)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    ws = (C.c_float * n)(*[float(w) for w in weights])
    _chk(lib().rdfc_combine_scalars(n, ptrs, ws, _f(post), _p(out), _st()), 'rdfc_combine_scalars')
    return out


def combine_scalars_backward(weights, post, grad_total, device):
    n = len(weights)
    out = torch.empty(n, dtype=torch.float32, device=device)
    ws = (C.c_float * n)(*[float(w) for w in weights])
    g = grad_total.reshape(()).float() if grad_total is not None else None
    _chk(lib().rdfc_combine_scalars_backward(n, ws, _f(post), _p(g), _p(out), _st()), 'rdfc_combine_scalars_backward')
    return out


def apply_pepper(img, mask_u8):
    N, Cc, H, W = img.shape
    _chk(lib().rdfc_apply_pepper(_p(img), _p(mask_u8), N, Cc, H * W, _st()), 'rdfc_apply_pepper')
    return img


# --------------------------------------------------------------------------- modulated deformable conv
def _md_desc(inp, weight, stride, pad, dil, group, dgroup):
    B, Cc, H, W = inp.shape
    Co, _, KH, KW = weight.shape
    return MDConvDesc(B, Cc, H, W, Co, KH, KW, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], group, dgroup)


def mdconv_forward(inp, offset, mask, weight, bias, stride, pad, dil, group, dgroup):
    d = _md_desc(inp, weight, stride, pad, dil, group, dgroup)
    Ho = (d.H + 2 * d.ph - (d.dh * (d.KH - 1) + 1)) // d.sh + 1
    Wo = (d.W + 2 * d.pw - (d.dw * (d.KW - 1) + 1)) // d.sw + 1
    out = torch.empty(d.B, d.Co, Ho, Wo, dtype=inp.dtype, device=inp.device)
    fn = lib().rdfc_mdconv_forward_f64 if inp.dtype == torch.float64 else lib().rdfc_mdconv_forward
    _chk(fn(C.byref(d), _p(inp), _p(offset), _p(mask), _p(weight), _p(bias), _p(out), _st()), 'rdfc_mdconv_forward')
    return out


def mdconv_backward(inp, offset, mask, weight, grad_out, stride, pad, dil, group, dgroup,
                    need=(True, True, True, True, True)):
    d = _md_desc(inp, weight, stride, pad, dil, group, dgroup)
    gi = torch.empty_like(inp) if need[0] else None
    go = torch.empty_like(offset) if need[1] else None
    gm = torch.empty_like(mask) if need[2] else None
    gw = torch.empty_like(weight) if need[3] else None
    gb = torch.empty(d.Co, dtype=inp.dtype, device=inp.device) if need[4] else None
    fn = lib().rdfc_mdconv_backward_f64 if inp.dtype == torch.float64 else lib().rdfc_mdconv_backward
    _chk(fn(C.byref(d), _p(inp), _p(offset), _p(mask), _p(weight), _p(grad_out), _p(gi), _p(go), _p(gm), _p(gw), _p(gb), _st()),
         'rdfc_mdconv_backward')
    return gi, go, gm, gw, gb


# --------------------------------------------------------------------------- NLSPN (nlspn.hip)
AFFINITY = dict(AS=0, ASS=1, TC=2, TGASS=3)


def nlspn_affinity(offset_aff, confidence, scale, w_conf, b_conf, affinity):
    """offset_aff: pixel-major [B,H,W,24] (f32 | compute dtype); confidence [B,1,H,W] f32 or None -> offset [B,18,H,W],
    aff [B,9,H,W], conf_samples [B,8,H,W] (or None)"""
    B, H, W, Cc = offset_aff.shape
    assert Cc == 24, 'the fused NLSPN kernels cover prop_kernel 3 (8 neighbours)'
    dev = offset_aff.device
    offset = torch.empty(B, 18, H, W, dtype=torch.float32, device=dev)
    aff = torch.empty(B, 9, H, W, dtype=torch.float32, device=dev)
    cs = torch.empty(B, 8, H, W, dtype=torch.float32, device=dev) if confidence is not None else None
    if confidence is not None:
        confidence = confidence.contiguous()
        assert confidence.dtype == torch.float32 and confidence.numel() == B * H * W
    _chk(lib().rdfc_nlspn_affinity_forward(_p(offset_aff), ld(offset_aff), dt(offset_aff), _p(confidence), _p(scale), _p(w_conf),
                                           _p(b_conf), AFFINITY[affinity], B, H, W, _p(offset), _p(aff), _p(cs), _st()),
         'rdfc_nlspn_affinity_forward')
    return offset, aff, cs


def nlspn_affinity_backward(offset_aff, conf_samples, scale, w_conf, affinity, grad_offset, grad_aff, want_conf, want_scale):
    B, H, W, _ = offset_aff.shape
    dev = offset_aff.device
    g_oa = torch.empty(B, H, W, 24, dtype=offset_aff.dtype, device=dev)
    g_conf = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) if want_conf and conf_samples is not None else None
    g_scale = torch.empty(1, dtype=torch.float32, device=dev) if want_scale else None
    go = grad_offset.contiguous() if grad_offset is not None else None
    _chk(lib().rdfc_nlspn_affinity_backward(_p(offset_aff), ld(offset_aff), dt(offset_aff), _p(conf_samples), _p(scale), _p(w_conf),
                                            AFFINITY[affinity], B, H, W, _p(go), _p(grad_aff.contiguous()), _p(g_oa), 24, _p(g_conf),
                                            _p(g_scale), _st()), 'rdfc_nlspn_affinity_backward')
    return g_oa, g_conf, g_scale


def nlspn_propagate(feat0, offset, aff, weight, bias, T):
    """-> feats [T+1,B,1,H,W] (slot 0 = feat0, slot T = the result)"""
    B, Cc, H, W = feat0.shape
    assert Cc == 1 and feat0.dtype == torch.float32
    feats = torch.empty(T + 1, B, 1, H, W, dtype=torch.float32, device=feat0.device)
    with _TimedHBM('prop_fwd', 116.0 * B * H * W * T, launches=T):          # per step and pixel: 18 offsets + 9 affinities + 1 read + 1 write, f32
        _chk(lib().rdfc_nlspn_propagate_forward(_p(feat0.contiguous()), _p(offset), _p(aff), _p(weight), _p(bias), B, H, W, T, _p(feats),
                                                _st()), 'rdfc_nlspn_propagate_forward')
    return feats


def nlspn_propagate_backward(feats, offset, aff, weight, grad_out, want_coeff):
    """-> grad_feat0 [B,1,H,W], grad_offset, grad_aff (or None, None)"""
    T = feats.shape[0] - 1
    _, B, _, H, W = feats.shape
    grads = torch.empty(T + 1, B, 1, H, W, dtype=torch.float32, device=feats.device)
    grads[T].copy_(grad_out)
    go = torch.empty_like(offset) if want_coeff else None
    gm = torch.empty_like(aff) if want_coeff else None
    _chk(lib().rdfc_nlspn_propagate_backward(_p(feats), _p(offset), _p(aff), _p(weight), B, H, W, T, _p(grads), _p(go), _p(gm), _st()),
         'rdfc_nlspn_propagate_backward')
    return grads[0], go, gm


# --------------------------------------------------------------------------- label net operators (label_ops.hip)
def maxpool2d(x, k, s, p):
    """pixel-major [N,H,W,C] -> (y [N,Ho,Wo,C], idx u8)"""
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty(N, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=x.device)
    _chk(lib().rdfc_maxpool2d_forward(_p(x), _p(y), _p(idx), N, H, W, Cc, ld(x), Cc, k, s, p, Ho, Wo, dt(x), _st()),
         'rdfc_maxpool2d_forward')
    return y, idx


def maxpool2d_backward(dy, idx, x_shape, k, s, p):
    N, H, W, Cc = x_shape
    Ho, Wo = dy.shape[1:3]
    dx = torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    _chk(lib().rdfc_maxpool2d_backward(_p(dy), _p(idx), _p(dx), N, H, W, Cc, ld(dy), Cc, k, s, p, Ho, Wo, dt(dy), _st()),
         'rdfc_maxpool2d_backward')
    return dx


def adaptive_avgpool(x, Ho, Wo):
    N, H, W, Cc = x.shape
    y = torch.empty(N, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    _chk(lib().rdfc_adaptive_avgpool_forward(_p(x), _p(y), N, H, W, Cc, ld(x), Cc, Ho, Wo, dt(x), _st()),
         'rdfc_adaptive_avgpool_forward')
    return y


def adaptive_avgpool_backward(dy, x_shape):
    N, H, W, Cc = x_shape
    Ho, Wo = dy.shape[1:3]
    dx = torch.empty(N, H, W, Cc, dtype=dy.dtype, device=dy.device)
    _chk(lib().rdfc_adaptive_avgpool_backward(_p(dy), _p(dx), N, H, W, Cc, ld(dy), Cc, Ho, Wo, dt(dy), _st()),
         'rdfc_adaptive_avgpool_backward')
    return dx


def bilinear(x, Ho, Wo, out=None, out_f32=False):
    """align_corners=True resize of a pixel-major map; `out` may be a channel window of a wider map (torch.cat target)"""
    N, H, W, Cc = x.shape
    if out is None:
        out = torch.empty(N, Ho, Wo, Cc, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    f32o = out.dtype == torch.float32 and x.dtype != torch.float32
    _chk(lib().rdfc_bilinear_forward(_p(x), _p(out), N, H, W, Cc, ld(x), ld(out), Ho, Wo, dt(x), int(f32o), _st()),
         'rdfc_bilinear_forward')
    return out


def bilinear_backward(dy, x_shape, x_dtype):
    N, H, W, Cc = x_shape
    Ho, Wo = dy.shape[1:3]
    dx = torch.empty(N, H, W, Cc, dtype=x_dtype, device=dy.device)
    f32y = dy.dtype == torch.float32 and x_dtype != torch.float32
    _chk(lib().rdfc_bilinear_backward(_p(dy), _p(dx), N, H, W, Cc, ld(dy), Cc, Ho, Wo, dt(dx), int(f32y), _st()),
         'rdfc_bilinear_backward')
    return dx


def scale_nc(x, scale):
    """x [N,H,W,C] * scale [N,C] (f32)"""
    N, H, W, Cc = x.shape
    y = torch.empty(N, H, W, Cc, dtype=x.dtype, device=x.device)
    _chk(lib().rdfc_scale_nc(_p(x), _p(scale), _p(y), N, H * W, Cc, ld(x), Cc, dt(x), _st()), 'rdfc_scale_nc')
    return y


def cross_entropy(logits, labels, ignore_index=-100):
    """logits pixel-major [N,H,W,C], labels i64 [N,H,W] -> (loss scalar, count scalar) on the device"""
    N, H, W, Cc = logits.shape
    P = N * H * W
    labels = labels.contiguous()
    assert labels.dtype == torch.int64 and labels.numel() == P
    partial = torch.empty(2 * lib().rdfc_cross_entropy_nchunk(P), dtype=torch.float64, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    _chk(lib().rdfc_cross_entropy(_p(logits), _p(labels), P, Cc, ld(logits), dt(logits), ignore_index, _p(partial), _p(out),
                                  out.data_ptr() + 4, _st()), 'rdfc_cross_entropy')
    return out[0], out[1]


def cross_entropy_backward(logits, labels, count, grad_out, ignore_index=-100, grad_scale=1.0):
    """grad_out: device scalar (the cotangent of the loss) or None"""
    N, H, W, Cc = logits.shape
    dl = torch.empty(N, H, W, Cc, dtype=logits.dtype, device=logits.device)
    if grad_out is not None:
        grad_out = grad_out.to(torch.float32).contiguous()
    _chk(lib().rdfc_cross_entropy_backward(_p(logits), _p(labels), N * H * W, Cc, ld(logits), dt(logits), ignore_index, _p(count),
                                           _p(grad_out), _f(grad_scale), _p(dl), Cc, _st()), 'rdfc_cross_entropy_backward')
    return dl


def _normal_loss_args(normal, gt_normal, norm_mask, logits_nchw):
    normal = normal.contiguous().float()
    N, _, H, W = normal.shape
    gt = gt_normal.contiguous().float() if gt_normal is not None else None
    m = None
    if norm_mask is not None and gt is not None:
        m = norm_mask.reshape(N, -1, H, W)[:, 0].contiguous().float()
    z, Cc, ldz = None, 0, 0
    if logits_nchw is not None:
        z = logits_nchw.detach().permute(0, 2, 3, 1)
        assert z.dtype == torch.float32 and z.shape[:3] == (N, H, W)
        Cc, ldz = z.shape[3], ld(z)
    return normal, gt, m, z, Cc, ldz, N, H * W


def normal_losses(normal, gt_normal, norm_mask, logits_nchw, label_wall, label_floor, label_ceiling, with_counts=False):
    """-> f32 [4] on the device: nor_loss, floor, wall, ceiling terms (unweighted); with_counts: [8], the four counts appended.
    logits_nchw: the label net's [N,C,H,W] channels-last view (or None); normal / gt_normal [N,3,H,W] f32"""
    normal, gt, m, z, Cc, ldz, N, HW = _normal_loss_args(normal, gt_normal, norm_mask, logits_nchw)
    partial = torch.empty(8 * lib().rdfc_normal_losses_nchunk(HW), dtype=torch.float64, device=normal.device)
    out = torch.empty(8, dtype=torch.float32, device=normal.device)
    _chk(lib().rdfc_normal_losses(_p(normal), _p(gt), _p(m), _p(z), N, HW, Cc, ldz, label_wall, label_floor, label_ceiling,
                                  _p(partial), _p(out), _st()), 'rdfc_normal_losses')
    return out if with_counts else out[:4]


def normal_losses_backward(normal, gt_normal, norm_mask, logits_nchw, label_wall, label_floor, label_ceiling, out8, grad4):
    normal, gt, m, z, Cc, ldz, N, HW = _normal_loss_args(normal, gt_normal, norm_mask, logits_nchw)
    g = torch.empty_like(normal)
    _chk(lib().rdfc_normal_losses_backward(_p(normal), _p(gt), _p(m), _p(z), N, HW, Cc, ldz, label_wall, label_floor, label_ceiling,
                                           _p(out8), _p(grad4.contiguous().float()), _p(g), _st()), 'rdfc_normal_losses_backward')
    return g

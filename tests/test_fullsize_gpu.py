"""GPU: the conv kernels at the benchmark's OWN launch sizes (16 stacked samples of BASELINE configs[1]: 228x304 ... 29x38 maps), through
properties that need no oracle run of that size (the oracle comparisons run at 60x76):
  * adjointness: <conv_w(x), g> = <x, dgrad_w(g)> = <w, wgrad(x, g)> -- the three kernels of a layer are each other's transposes, so a
    tile that is skipped, doubled or mis-addressed anywhere in a 1 000-workgroup grid shows up in one scalar;
  * linearity in the exactly representable case: conv(2 x) == 2 conv(x) bit for bit (a power of two commutes with every rounding);
  * reproducibility under load: the same launch repeated while another stream keeps every CU busy returns the same bits.  (The
    write-after-read hazard of the LDS-DMA rings found in round 3 -- common.h: wait_vm_and_reads -- is guarded by
    test_shapes_gpu.py's 480x640 run-to-run test, which fails on a -DRDFC_TEST_NO_READ_WAIT build; this synthetic pairing of
    launches did not provoke it on that build and stays as the cheap first line.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# N, H, W, Ci, Co, k, stride, pad: the step's big layers (rdfc_gan_amd/DESIGN.md section 7's per-layer table)
LAYERS = [(16, 228, 304, 64, 64, 3, 1, 1), (16, 114, 152, 128, 128, 3, 1, 1), (16, 57, 76, 256, 256, 3, 1, 1), (16, 29, 38, 512, 512, 3, 1, 1),
          (16, 228, 304, 128, 64, 3, 1, 1), (16, 228, 304, 64, 128, 3, 2, 1), (16, 114, 152, 128, 256, 3, 2, 1), (16, 28, 38, 256, 512, 4, 1, 1),
          (16, 57, 76, 128, 256, 4, 2, 1), (16, 57, 76, 384, 768, 1, 1, 0), (16, 228, 304, 128, 32, 3, 1, 1),
          # round 6: the 3x3 / stride-2 64 -> 192 stem of the depth branch and W-AdaIN's first 1x1 (round-5 verdict, item 1), NLSPN's guidance
          # path on the narrow-map kernels (conv_narrow.hip: conv_offset_aff 8 -> 24, gd_dec0 128 -> 8)
          (16, 228, 304, 64, 192, 3, 2, 1), (16, 114, 152, 192, 384, 1, 1, 0), (16, 228, 304, 8, 24, 3, 1, 1), (16, 228, 304, 128, 8, 3, 1, 1)]
IDS = ['x'.join(str(v) for v in c) for c in LAYERS]
DT = [torch.float16, torch.bfloat16]


def _layer(case, dtype, seed=0):
    from rdfc_gan_amd import hip
    N, H, W, Ci, Co, k, s, p = case
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(dtype)
    w = torch.randn(Co, Ci, k, k, device=DEV, generator=g) * (0.7 / (Ci * k * k) ** 0.5)
    w = w.to(dtype).float()                                  # the packed operand holds these values exactly
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gy = torch.randn(N, Ho, Wo, Co, device=DEV, generator=g).to(dtype)
    wf, wb = hip.pack_weight(w.contiguous(), dtype)
    d = hip.conv_desc(x.shape, Co, k, s, p, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    return hip, d, x, w, wf, wb, gy, (N, Ho, Wo, Co)


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', LAYERS, ids=IDS)
def test_forward_dgrad_wgrad_are_adjoint(case, dtype):
    hip, d, x, w, wf, wb, gy, oshape = _layer(case, dtype)
    y = hip.conv_forward(d, x, wf, None, torch.empty(oshape, dtype=dtype, device=DEV))
    dx = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x))
    dw = hip.conv_wgrad(d, x, gy)                                                   # f32 [Co][k*k][Ci]
    a = (y.double() * gy.double()).sum()
    b = (x.double() * dx.double()).sum()
    c = (dw.double() * w.permute(0, 2, 3, 1).reshape(dw.shape).double()).sum()
    # y and dx are stored in 16 bits (relative rounding 2^-9 bf16 / 2^-11 f16 per element, independent signs): the scale of what
    # that does to the sums is the root-sum-square of their terms
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    sa = float((y.double() * gy.double()).pow(2).sum().sqrt())
    sb = float((x.double() * dx.double()).pow(2).sum().sqrt())
    assert abs(float(a - c)) <= 6 * eps * sa + 1e-6 * abs(float(c)), (float(a), float(c), sa)
    assert abs(float(b - c)) <= 6 * eps * sb + 1e-6 * abs(float(c)), (float(b), float(c), sb)


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', LAYERS, ids=IDS)
def test_values_against_torch_fp32_on_the_host(case, dtype):
    """forward, input gradient and weight gradient of the step's largest launch shapes against torch.nn.functional.conv2d in fp32 on the
    host, on identically rounded operands (round-4 verdict: the value comparisons of test_kernels_gpu.py stop at 130x140x64 maps; here the
    launches are the benchmark's own: 16 stacked samples).  Limits: test_kernels_gpu.py's tol(dtype) -- the rounding of the stored result
    relative to the largest reference value."""
    import torch.nn.functional as F
    hip, d, x, w, wf, wb, gy, oshape = _layer(case, dtype)
    N, H, W, Ci, Co, k, s, p = case
    tol = {torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dtype]
    y = hip.conv_forward(d, x, wf, None, torch.empty(oshape, dtype=dtype, device=DEV))
    dx = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x))
    dw = hip.unpack_wgrad(hip.conv_wgrad(d, x, gy), Co, Ci, k, k)
    xr = x.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)           # (channels-last strides: what the host library likes)
    wr = w.cpu().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, p)
    y_ref.backward(gy.float().cpu().permute(0, 3, 1, 2))

    def rel(got, ref):
        return float((got.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-12))
    e_y = rel(y.float().cpu().permute(0, 3, 1, 2), y_ref.detach())
    e_dx = rel(dx.float().cpu().permute(0, 3, 1, 2), xr.grad)
    e_dw = rel(dw.cpu(), wr.grad)
    print(f'[full-size values {"x".join(map(str, case))} {dtype}] forward {e_y:.2e} dgrad {e_dx:.2e} wgrad {e_dw:.2e} (limit {tol:.1e}; wgrad: f32 result, 1e-3)')
    assert e_y < tol and e_dx < tol, (e_y, e_dx)
    assert e_dw < 1e-3, e_dw                                                 # f32 result of exact products: only the order of the f32 adds differs


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', LAYERS, ids=IDS)
def test_scaling_by_two_is_exact(case, dtype):
    hip, d, x, w, wf, wb, gy, oshape = _layer(case, dtype, seed=1)
    def same(two, one):
        # (f16 results below 2^-13 are subnormal after the doubling or before it: fixed absolute spacing, 2 round(a) != round(2a))
        ok = one.abs() >= 2.0 ** -13 if dtype == torch.float16 else torch.ones_like(one, dtype=torch.bool)
        assert float(ok.float().mean()) > 0.99
        return torch.equal(two[ok], (one * 2)[ok])
    y1 = hip.conv_forward(d, x, wf, None, torch.empty(oshape, dtype=dtype, device=DEV))
    y2 = hip.conv_forward(d, x * 2, wf, None, torch.empty(oshape, dtype=dtype, device=DEV))
    assert same(y2, y1)
    dx1 = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x))
    dx2 = hip.conv_dgrad(d, gy * 2, wb, None, torch.empty_like(x))
    assert same(dx2, dx1)
    dw1 = hip.conv_wgrad(d, x, gy)
    dw2 = hip.conv_wgrad(d, x * 2, gy)
    if hip.wgrad_is_deterministic(d):
        assert torch.equal(dw2, dw1 * 2)
    else:                                                    # split-K float atomics: the order of the adds varies
        assert float((dw2 - dw1 * 2).abs().max()) <= 1e-5 * float(dw1.abs().max())


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', LAYERS[:6], ids=IDS[:6])
def test_results_do_not_change_under_load(case, dtype):
    """forward, input gradient and weight gradient launched on one stream while a second stream runs other LDS-DMA convolutions
    back to back (every CU's LDS queue shared by two kernels): twelve rounds, the same bits as alone"""
    hip, d, x, w, wf, wb, gy, oshape = _layer(case, dtype, seed=2)
    _, d2, x2, _w2, wf2, wb2, gy2, oshape2 = _layer(LAYERS[0] if case != LAYERS[0] else LAYERS[2], dtype, seed=3)
    y_ref = hip.conv_forward(d, x, wf, None, torch.empty(oshape, dtype=dtype, device=DEV)).clone()
    dx_ref = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x)).clone()
    dw_ref = hip.conv_wgrad(d, x, gy).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    y2, dx2 = torch.empty(oshape2, dtype=dtype, device=DEV), torch.empty_like(x2)
    for r in range(12):
        with torch.cuda.stream(side):
            for _ in range(6):
                hip.conv_forward(d2, x2, wf2, None, y2)
                hip.conv_dgrad(d2, gy2, wb2, None, dx2)
        y = hip.conv_forward(d, x, wf, None, torch.full(oshape, 7.0, dtype=dtype, device=DEV))
        dx = hip.conv_dgrad(d, gy, wb, None, torch.full_like(x, 7.0))
        dw = hip.conv_wgrad(d, x, gy)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), (r, 'forward', float((y.float() - y_ref.float()).abs().max()))
        assert torch.equal(dx, dx_ref), (r, 'dgrad', float((dx.float() - dx_ref.float()).abs().max()))
        if hip.wgrad_is_deterministic(d):
            assert torch.equal(dw, dw_ref), (r, 'wgrad', float((dw - dw_ref).abs().max()))


HEIGHT_CASES = [(16, 57, 76, 256, 256), (16, 114, 152, 128, 128), (16, 29, 38, 512, 512), (16, 228, 304, 64, 128), (6, 97, 131, 192, 320)]


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', HEIGHT_CASES, ids=['x'.join(str(v) for v in c) for c in HEIGHT_CASES])
def test_every_tile_height_gives_the_same_bits(case, dtype):
    """round 5: the 128-channel 3x3 kernel picks its pixel-tile height (4 / 6 / 8 / 10 rows of 16 per wave) per launch; every height walks the
    taps in the same order, so forward (bias + activation), fused statistics and input gradient (with an addend) must not depend on it."""
    from rdfc_gan_amd import hip
    N, H, W, Ci, Co = case
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(dtype)
    w = torch.randn(Co, Ci, 3, 3, device=DEV, generator=g) * (0.7 / (Ci * 9) ** 0.5)
    b = torch.randn(Co, device=DEV, generator=g) * 0.1
    gy = torch.randn(N, H, W, Co, device=DEV, generator=g).to(dtype)
    add = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(dtype)
    wf, wb = hip.pack_weight(w.contiguous(), dtype)
    d = hip.conv_desc(x.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=hip.dt(x), act='lrelu', slope=0.2)
    d0 = hip.conv_desc(x.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    lib = hip.lib()
    ref, ref_stats, names = None, None, set()
    try:
        for jr in (4, 6, 8, 10):
            stats = None
            lib.rdfc_tune_glds_rows(jr)
            y = torch.full((N, H, W, Co), 3.0, dtype=dtype, device=DEV)
            hip.conv_forward(d, x, wf, b, y)
            names.add(lib.rdfc_last_kernel().decode())
            out = [y]
            if hip.conv_stats_nchunk(d0, 1) > 0:
                y2 = torch.full((N, H, W, Co), 3.0, dtype=dtype, device=DEV)
                mean, rstd = hip.conv_forward_stats(d0, x, wf, b, y2, 1, 1e-5)
                out += [y2]                                     # (the statistics themselves: a different tiling adds the partial rows in a different order -- compared to 1e-6 below)
                stats = (mean.clone(), rstd.clone())
            dx = torch.full_like(x, 3.0)
            hip.conv_dgrad(d0, gy, wb, None, dx, addend=add)
            out.append(dx)
            if ref is None:
                ref, ref_stats = out, stats
            else:
                for i, (a, r) in enumerate(zip(out, ref)):
                    assert torch.equal(a, r), (jr, i, float((a.float() - r.float()).abs().max()))
                for a, r in zip(stats or (), ref_stats or ()):
                    assert float((a - r).abs().max() / r.abs().max()) < 1e-6
    finally:
        lib.rdfc_tune_glds_rows(0)
    assert len(names) == 4, names                               # four different instantiations ran


def test_igemm_glds_takes_batches_beyond_2_21_pixels():
    """round-5 advice: the LDS-DMA implicit-GEMM kernel divided a flat pixel index with a float reciprocal that is exact below 2^21 and REFUSED
    larger launches (batch 8 at 480 x 640, batch 32 at 228 x 304: a silent fall back to the register-staged kernel).  Round 6 divides the
    batch-wide index through a double reciprocal (conv_igemm_glds.hip: qdiv_wide): a 1x1 192 -> 64 layer over 8 x 480 x 640 = 2.46 M pixels
    must run on igemm_glds and give the values of a host matmul on sampled pixels, the last image's last row included."""
    from rdfc_gan_amd import hip
    N, H, W, Ci, Co = 8, 480, 640, 192, 64
    dtype = torch.float16
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(dtype)
    w = (torch.randn(Co, Ci, 1, 1, device=DEV, generator=g) * (0.7 / Ci ** 0.5)).to(dtype).float()
    wf, _wb = hip.pack_weight(w.contiguous(), dtype)
    d = hip.conv_desc(x.shape, Co, 1, 1, 0, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    y = hip.conv_forward(d, x, wf, None, torch.empty(N, H, W, Co, dtype=dtype, device=DEV))
    assert 'igemm_glds_kernel' in hip.lib().rdfc_last_kernel().decode(), hip.lib().rdfc_last_kernel().decode()
    P = N * H * W
    idx = torch.cat([torch.randint(0, P, (8192,), generator=torch.Generator().manual_seed(6)), torch.arange(P - 640, P), torch.arange((1 << 21) - 64, (1 << 21) + 64)])
    ref = x.view(P, Ci)[idx.to(DEV)].float().cpu() @ w.view(Co, Ci).cpu().t()
    got = y.view(P, Co)[idx.to(DEV)].float().cpu()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-3


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', [LAYERS[0], LAYERS[1], LAYERS[5], LAYERS[6], LAYERS[8]], ids=[IDS[0], IDS[1], IDS[5], IDS[6], IDS[8]])
def test_epilogue_addends_in_whole_rows_give_the_same_bits(case, dtype):
    """round 6: rdfc_conv2d_dgrad_add / _add2 at the benchmark's launch sizes -- the addends fetched as 16-byte pieces by the epilogue's flush
    from an f32 staging tile (common.h: rsf_flush; 16-byte aligned maps) against the 8-byte pieces on the way into the 16-bit tile (maps
    that are only 8-byte aligned: channel slices starting at channel 4), which is the path every earlier round's parity tests ran on.  The
    arithmetic and its order are the same: bit-equal.  And both equal the plain input gradient + the maps within one rounding."""
    hip, d, x, w, wf, wb, gy, oshape = _layer(case, dtype, seed=2)
    N, H, W, Ci, Co, k, s, p = case
    g = torch.Generator(device=DEV).manual_seed(9)
    wide = [torch.randn(N, H, W, Ci + 8, device=DEV, generator=g).to(dtype) for _ in range(2)]
    al = [t[..., 8:] for t in wide]                       # 16-byte aligned slices
    odd = []
    for t in al:
        o = torch.zeros(N, H, W, Ci + 8, dtype=dtype, device=DEV)[..., 4:4 + Ci]
        o.copy_(t)
        odd.append(o)
    assert al[0].data_ptr() % 16 == 0 and odd[0].data_ptr() % 16 == 8
    plain = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x))
    one_a = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x), addend=al[0])
    one_o = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x), addend=odd[0])
    assert torch.equal(one_a, one_o)
    ref = plain.float() + al[0].float()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((one_a.float() - ref).abs().max()) <= 2 * eps * float(ref.abs().max())
    if s == 2:                                            # the transposed stride-2 kernel takes two
        two_a = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x), addend=al[0], addend2=al[1])
        two_o = hip.conv_dgrad(d, gy, wb, None, torch.empty_like(x), addend=odd[0], addend2=odd[1])
        assert torch.equal(two_a, two_o)


@pytest.mark.parametrize('dtype', DT, ids=['f16', 'bf16'])
@pytest.mark.parametrize('case', [LAYERS[0], LAYERS[1], LAYERS[4], LAYERS[10]], ids=[IDS[0], IDS[1], IDS[4], IDS[10]])
def test_three_kernel_rows_per_workgroup_give_the_same_bits(case, dtype):
    """round 6: the ring weight gradient's 64-co tile runs the three kernel rows of a tile as one 768-thread workgroup (three four-wave
    groups in lockstep, conv_wgrad_ring.hip: KG) instead of three 256-thread workgroups: same slices, same walk, same order of the f32
    adds -- bit-equal to the rounds 3-5 launch on the 64-channel full-resolution layers, the 128 -> 128 and the 128 -> 32 ones."""
    hip, d, x, w, wf, wb, gy, oshape = _layer(case, dtype, seed=4)
    lib = hip.lib()
    try:
        lib.rdfc_tune_wgrad_kg(1)
        a = hip.conv_wgrad(d, x, gy)
        ka = lib.rdfc_last_kernel().decode()
        lib.rdfc_tune_wgrad_kg(0)
        b = hip.conv_wgrad(d, x, gy)
        kb = lib.rdfc_last_kernel().decode()
    finally:
        lib.rdfc_tune_wgrad_kg(1)
    assert ka.endswith(', 3>') and kb.endswith(', 1>') and 'wgrad_ring2_kernel<64' in ka, (ka, kb)
    assert torch.equal(a, b)

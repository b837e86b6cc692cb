"""A/B of the ring weight-gradient kernel (conv_wgrad_ring.hip: wgrad_ring2_kernel, mode 3 = every launch it can take) against the
tap-gather / segment-walk kernels with float atomics (mode 0) on the step's layers: interleaved rounds in ONE process (the mode is
switched between launches through rdfc_tune_wgrad_ring), random operands; the ring kernel's result must repeat bit for bit
(fixed-order slab reduction) and agree with the other kernel to f32 summation-order accuracy.  (Round 5's MODE=ring2 leg -- round 3's
wgrad_ring_kernel against wgrad_ring2_kernel, bit-equal on all shapes: profiles/r05_wgrad_ring2_ab.txt -- went with the old kernel.)
python tools/wgrad_ring_bench.py [rounds] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [  # N, H, W, Ci, Co, pad, reflect, kernel, stride
    (16, 57, 76, 256, 256, 1, 0, 3, 1), (16, 29, 38, 512, 512, 1, 0, 3, 1), (16, 114, 152, 128, 128, 1, 0, 3, 1), (16, 228, 304, 64, 64, 1, 0, 3, 1),
    (16, 228, 304, 128, 64, 1, 0, 3, 1), (16, 114, 152, 256, 128, 1, 0, 3, 1), (16, 57, 76, 256, 256, 1, 1, 3, 1), (3, 29, 38, 128, 192, 1, 0, 3, 1),
    (2, 31, 45, 64, 128, 0, 0, 3, 1), (16, 228, 304, 128, 32, 1, 0, 3, 1), (2, 40, 52, 128, 32, 1, 0, 3, 1),
    # stride 2 / 4x4 (the encoders' down-sampling convs, the transposed decoder convs seen from their output, PatchGAN)
    (16, 114, 152, 128, 256, 1, 0, 3, 2), (16, 228, 304, 64, 128, 1, 0, 3, 2), (16, 228, 304, 64, 192, 1, 0, 3, 2), (16, 57, 76, 256, 512, 1, 0, 3, 2),
    (16, 58, 76, 128, 768, 1, 0, 3, 2), (16, 114, 152, 64, 384, 1, 0, 3, 2), (16, 114, 152, 64, 128, 1, 0, 4, 2), (16, 57, 76, 128, 256, 1, 0, 4, 2),
    (16, 28, 38, 256, 512, 1, 0, 4, 1), (3, 31, 61, 64, 128, 1, 0, 4, 2), (2, 33, 59, 128, 64, 1, 0, 3, 2)]
if len(sys.argv) > 3:
    SHAPES = [sh for sh in SHAPES if (sh[7], sh[8]) == tuple(int(v) for v in sys.argv[3].split(','))]
VAR = 'RDFC_WGRAD_RING'
MODES = ('0', '3')
dtn = os.environ.get('PP_DTYPE', 'bf16')
dt = dict(bf16=torch.bfloat16, f16=torch.float16)[dtn]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (N, H, W, Ci, Co, pad, reflect, ksz, st) in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(1)
    Ho, Wo = (H + 2 * pad - ksz) // st + 1, (W + 2 * pad - ksz) // st + 1
    x = torch.randn(N, H, W, Ci, device='cuda', generator=g).to(dt)
    gy = torch.randn(N, Ho, Wo, Co, device='cuda', generator=g).to(dt)
    d = hip.conv_desc(x.shape, Co, ksz, st, pad, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    outs, kern = {}, {}
    for m in MODES:
        hip.lib().rdfc_tune_wgrad_ring(int(m))
        bad = 0
        for rep in range(4):
            dw = hip.conv_wgrad(d, x, gy, reflect=bool(reflect))
            kern[m] = hip.lib().rdfc_last_kernel().decode()
            if m not in outs:
                outs[m] = dw
            elif not torch.equal(outs[m], dw):
                bad += 1
        print(f'  {VAR}={m} ({kern[m]}): {bad} of 3 repeats differ from the first run')
    a, b = outs[MODES[0]].double(), outs[MODES[1]].double()
    rel = float((a - b).abs().max() / a.abs().max())
    print(f'wgrad N{N} {H}x{W} {Ci}->{Co} k{ksz}s{st} pad{pad} reflect{reflect} [{dtn}]: max|{MODES[1]} - {MODES[0]}| / max = {rel:.2e} nan={int(torch.isnan(b).sum())} bit-equal={torch.equal(outs[MODES[0]], outs[MODES[1]])}', flush=True)
    fl = 2.0 * N * Ho * Wo * Co * Ci * ksz * ksz
    buf = torch.zeros(Co, ksz * ksz, Ci, device='cuda')
    res = {m: [] for m in MODES}
    for r in range(rounds):
        for m in MODES:
            hip.lib().rdfc_tune_wgrad_ring(int(m))
            res[m].append(timed(lambda: hip.conv_wgrad(d, x, gy, into=buf, reflect=bool(reflect))))
    line = '  '
    for m in MODES:
        v = sorted(res[m])
        med = v[len(v) // 2]
        line += f'  {VAR[5:]}={m}: median {med:7.1f} us (min {v[0]:7.1f}) {fl / med / 1e6:6.0f} TFLOP/s'
    print(line, flush=True)

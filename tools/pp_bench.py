"""A/B of the ping-pong 3x3 kernel (conv_pp.hip, RDFC_PP=1|2) against conv_glds.hip (RDFC_PP=0) on the step's 3x3 / stride-1 layers:
interleaved rounds in ONE process (the switch is read per launch), random operands, outputs compared bit for bit (both kernels add
the taps and slabs in the same order, so any difference is a race).  python tools/pp_bench.py [rounds] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [  # N, H, W, Ci, Co (3x3, stride 1, pad 1): the batched step's launches
    (16, 57, 76, 256, 256), (16, 29, 38, 512, 512), (16, 114, 152, 128, 128), (16, 228, 304, 64, 64), (16, 228, 304, 128, 64),
    (16, 114, 152, 256, 128), (16, 59, 78, 256, 256), (8, 57, 76, 256, 256), (3, 29, 38, 128, 192)]
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


for (N, H, W, Ci, Co) in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, H, W, Ci, device='cuda', generator=g).to(dt)
    gy = torch.randn(N, H, W, Co, device='cuda', generator=g).to(dt)
    w = torch.randn(Co, Ci, 3, 3, device='cuda', generator=g) * (Ci * 9) ** -0.5
    bias = torch.randn(Co, device='cuda', generator=g)
    wf, wb = hip.pack_weight(w, dt)
    d = hip.conv_desc(x.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=hip.dt(x), act='lrelu', slope=0.2)
    dd = hip.conv_desc(x.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    modes = ['0', '1'] if Co > 64 else ['0', '2']
    outs, kern = {}, {}
    for which in ('fwd', 'dgrad'):
        for m in modes:
            os.environ['RDFC_PP'] = m
            bad = 0
            for rep in range(4):
                if which == 'fwd':
                    y = torch.empty(N, H, W, Co, device='cuda', dtype=dt)
                    hip.conv_forward(d, x, wf, bias, y)
                else:
                    y = torch.empty(N, H, W, Ci, device='cuda', dtype=dt)
                    hip.conv_dgrad(dd, gy, wb, None, y)
                kern[(which, m)] = hip.lib().rdfc_last_kernel().decode()
                if (which, m) not in outs:
                    outs[(which, m)] = y
                elif not torch.equal(outs[(which, m)], y):
                    bad += 1
            if bad:
                print(f'  !! {which} RDFC_PP={m}: {bad} of 3 repeats differ from the first run (race)')
        a, b = outs[(which, modes[0])], outs[(which, modes[1])]
        same = torch.equal(a, b)
        diff = float((a.float() - b.float()).abs().max())
        nan = int(torch.isnan(b.float()).sum())
        print(f'{which} N{N} {H}x{W} {Ci}->{Co} [{dtn}]: {kern[(which, modes[1])]} vs {kern[(which, modes[0])]}: bit-equal={same} max|diff|={diff:.3e} nan={nan}', flush=True)
    fl = 2.0 * N * H * W * Co * Ci * 9
    ybuf = torch.empty(N, H, W, Co, device='cuda', dtype=dt)
    xbuf = torch.empty(N, H, W, Ci, device='cuda', dtype=dt)
    res = {(wh, m): [] for wh in ('fwd', 'dgrad') for m in modes}
    for r in range(rounds):
        for wh in ('fwd', 'dgrad'):
            for m in modes:
                os.environ['RDFC_PP'] = m
                fn = (lambda: hip.conv_forward(d, x, wf, bias, ybuf)) if wh == 'fwd' else (lambda: hip.conv_dgrad(dd, gy, wb, None, xbuf))
                res[(wh, m)].append(timed(fn))
    for wh in ('fwd', 'dgrad'):
        line = f'  {wh}:'
        for m in modes:
            v = sorted(res[(wh, m)])
            med = v[len(v) // 2]
            line += f'  PP={m}: median {med:7.1f} us (min {v[0]:7.1f}) {fl / med / 1e6:6.0f} TFLOP/s'
        print(line, flush=True)

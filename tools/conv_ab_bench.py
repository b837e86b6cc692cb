"""A/B of two builds of the library on a list of conv launches (forward / dgrad through the C ABI): one process per build (RDFC_HIP_LIB),
interleaved by the caller; HIP events around batches of launches, f16, random operands.
usage: RDFC_HIP_LIB=<lib> python tools/conv_ab_bench.py [rounds] [launches per timing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = torch.float16
SHAPES = [  # N, H, W, Ci, Co, k, s, p, which   -- the step's igemm_glds / convt3x3s2_glds launches (profiles/r04_by_shape_one_stream.txt)
    (16, 28, 38, 256, 512, 4, 1, 1, 'fwd'), (8, 28, 38, 256, 512, 4, 1, 1, 'fwd'), (16, 228, 304, 64, 128, 3, 2, 1, 'fwd'),
    (16, 114, 152, 128, 256, 3, 2, 1, 'fwd'), (16, 228, 304, 64, 192, 3, 2, 1, 'fwd'), (16, 114, 152, 192, 384, 1, 1, 0, 'fwd'),
    (16, 57, 76, 384, 768, 1, 1, 0, 'fwd'), (16, 57, 76, 256, 512, 3, 2, 1, 'fwd'), (16, 29, 38, 768, 1536, 1, 1, 0, 'fwd'),
    (16, 58, 76, 128, 768, 3, 2, 1, 'fwd'), (16, 114, 152, 64, 384, 3, 2, 1, 'fwd'), (16, 57, 76, 128, 256, 4, 2, 1, 'fwd'),
    (16, 228, 304, 64, 128, 3, 2, 1, 'dgrad'), (16, 114, 152, 128, 256, 3, 2, 1, 'dgrad'), (16, 57, 76, 256, 512, 3, 2, 1, 'dgrad'),
    (16, 228, 304, 64, 192, 3, 2, 1, 'dgrad'), (16, 114, 152, 64, 384, 3, 2, 1, 'dgrad'), (16, 114, 152, 64, 128, 4, 2, 1, 'dgrad'),
    (16, 28, 38, 256, 512, 4, 1, 1, 'dgrad'), (16, 114, 152, 192, 384, 1, 1, 0, 'dgrad'),
]
if os.environ.get('AB_SHAPES'):                     # e.g. AB_SHAPES='[(16,228,304,64,64,3,1,1,"fwd")]'
    SHAPES = eval(os.environ['AB_SHAPES'])
for (N, H, W, Ci, Co, k, s, p, which) in SHAPES:
    x = torch.randn(N, H, W, Ci, device='cuda').to(dt)
    w = torch.randn(Co, Ci, k, k, device='cuda') * 0.05
    wf, wb = hip.pack_weight(w, dt)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty(N, Ho, Wo, Co, device='cuda', dtype=dt)
    gy = torch.randn(N, Ho, Wo, Co, device='cuda').to(dt)
    dx = torch.empty_like(x)
    d = hip.conv_desc(x.shape, Co, k, s, p, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    run = (lambda: hip.conv_forward(d, x, wf, None, y)) if which == 'fwd' else (lambda: hip.conv_dgrad(d, gy, wb, None, dx))
    fl = 2.0 * N * Ho * Wo * Co * Ci * k * k
    ts = []
    run(); kern = hip.lib().rdfc_last_kernel().decode()
    for r in range(rounds):
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    cs = float((y if which == 'fwd' else dx).float().abs().sum())
    print(f'{which:5s} {N}x{H}x{W} {Ci}->{Co} k{k}s{s}: {ts[len(ts) // 2]:7.1f} us (min {ts[0]:6.1f}) {fl / ts[len(ts) // 2] / 1e6:6.0f} TF  {kern}  checksum {cs:.6e}', flush=True)

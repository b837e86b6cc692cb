"""A/B of the pixel-tile heights of conv3x3_glds (rdfc_tune_glds_rows) on the step's 128-co 3x3 / stride-1 launch shapes: interleaved
rounds in ONE process (cdna_hip_programming.md rule 24), HIP events around batches of launches, f16, random operands.
usage: python tools/conv_tall_bench.py [rounds] [launches per timing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
JRS = [int(v) for v in os.environ.get('JRS', '4,6,8').split(',')]
dt = torch.float16
SHAPES = [  # N, H, W, Ci, Co, pad, which
    (16, 57, 76, 256, 256, 1, 'fwd'), (16, 57, 76, 256, 256, 1, 'dgrad'), (16, 59, 78, 256, 256, 0, 'fwd'),
    (16, 114, 152, 128, 128, 1, 'fwd'), (16, 29, 38, 512, 512, 1, 'fwd'), (16, 228, 304, 64, 128, 1, 'fwd'),
    (8, 120, 160, 256, 256, 1, 'fwd'), (8, 60, 80, 256, 256, 1, 'fwd'),      # every tile full, many rounds: the steady-state rate
    (4, 120, 160, 128, 128, 1, 'fwd'), (4, 240, 320, 128, 128, 1, 'fwd'),     # configuration #5's maps (480 x 640 inputs)
]
if os.environ.get('SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['SHAPES'].split(',')]
lib = hip.lib()
for (N, H, W, Ci, Co, p, which) in SHAPES:
    x = torch.randn(N, H, W, Ci, device='cuda').to(dt)
    w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05
    wf, wb = hip.pack_weight(w, dt)
    Ho, Wo = H + 2 * p - 2, W + 2 * p - 2
    y = torch.empty(N, Ho, Wo, Co, device='cuda', dtype=dt)
    gy = torch.randn(N, Ho, Wo, Co, device='cuda').to(dt)
    dx = torch.empty_like(x)
    d = hip.conv_desc(x.shape, Co, 3, 1, p, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    run = (lambda: hip.conv_forward(d, x, wf, None, y)) if which == 'fwd' else (lambda: hip.conv_dgrad(d, gy, wb, None, dx))
    fl = 2.0 * N * Ho * Wo * Co * Ci * 9
    ref, times, names = None, {j: [] for j in JRS}, {}
    for j in JRS:                                    # same bits whatever the tile height
        lib.rdfc_tune_glds_rows(j)
        run()
        names[j] = lib.rdfc_last_kernel().decode()
        out = (y if which == 'fwd' else dx).clone()
        if ref is None: ref = out
        elif not torch.equal(ref, out): print(f'  !! JR={j} differs from JR={JRS[0]}: max {float((ref.float() - out.float()).abs().max()):.3e}')
    for r in range(rounds):
        for j in JRS:
            lib.rdfc_tune_glds_rows(j)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): run()
            e1.record(); e1.synchronize()
            times[j].append(e0.elapsed_time(e1) * 1e3 / reps)
    lib.rdfc_tune_glds_rows(0)
    run(); auto = lib.rdfc_last_kernel().decode()
    line = ' | '.join(f'JR{j}: {sorted(times[j])[len(times[j]) // 2]:7.1f} us (min {min(times[j]):6.1f}) {fl / sorted(times[j])[len(times[j]) // 2] / 1e6:6.0f} TF' for j in JRS)
    print(f'{which:5s} {N}x{H}x{W} {Ci}->{Co} p{p}: {line} | auto -> {auto}', flush=True)

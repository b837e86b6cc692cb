"""Reads the s_memtime stamps of the diagnostic conv_glds build (tools/glds_stamps.sh) and prints, per wave, the average
cycles a phase spends in: vmcnt wait | barrier | DMA issue | fragment-read issue | MFMA issue (until the last MFMA is issued)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rdfc_gan_amd import hip
N, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
dt = torch.bfloat16
x = torch.randn(N, H, W, Ci, device='cuda').to(dt)
w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05
wf, wb = hip.pack_weight(w, dt)
y = torch.empty(N, H, W, Co, device='cuda', dtype=dt)
d = hip.conv_desc(x.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=hip.dt(x))
for _ in range(20):
    hip.conv_forward(d, x, wf, None, y)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 9 * 6))()
fn = hip.lib().rdfc_debug_glds_stamps
fn.argtypes = [C.c_void_p]
assert fn(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(8, 9, 6).astype(np.int64)
names = ['vmcnt wait', 'barrier', 'DMA issue', 'frag reads (issue + data back)', 'MFMA issue']
print(f'shape N{N} {H}x{W} {Ci}->{Co}; stamps of the workgroup in the middle of the grid, slab 2 (memtime ticks = shader cycles)')
for wv in range(8):
    if s[wv].max() == 0:
        continue
    seg = np.diff(s[wv], axis=1)                    # [9 taps][5 segments]
    gap = s[wv][1:, 0] - s[wv][:-1, 5]              # end of a phase -> start of the next
    phase = s[wv][1:, 0] - s[wv][:-1, 0]
    print(f'wave {wv}: phase {phase.mean():6.0f} cyc | ' + ' | '.join(f'{n} {seg[:, i].mean():5.0f}' for i, n in enumerate(names)) + f' | between phases {gap.mean():4.0f}')
    print('        per tap phase lengths:', phase.tolist())

"""Tile-level timeline of conv3x3_glds (diagnostic build, tools/glds_stamps.sh ... tile): s_memtime stamps of ~512 sampled workgroups at
entry | operands of phase 0 landed | K loop done | stores issued.  Prints the mean cycles of the three parts, and how many sampled
workgroups were alive at the same time (the launch's span / a tile's life).  usage: python tools/glds_tile_stamps.py N H W Ci Co [fwd|dgrad] [JR]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rdfc_gan_amd import hip
N, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
which = sys.argv[6] if len(sys.argv) > 6 else 'fwd'
jr = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dt = torch.bfloat16
x = torch.randn(N, H, W, Ci, device='cuda').to(dt)
w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05
wf, wb = hip.pack_weight(w, dt)
y = torch.empty(N, H, W, Co, device='cuda', dtype=dt)
gy = torch.randn(N, H, W, Co, device='cuda').to(dt)
dx = torch.empty_like(x)
d = hip.conv_desc(x.shape, Co, 3, 1, 1, ldx=Ci, ldy=Co, dtype=hip.dt(x))
lib = hip.lib()
lib.rdfc_tune_glds_rows(jr)
run = (lambda: hip.conv_forward(d, x, wf, None, y)) if which == 'fwd' else (lambda: hip.conv_dgrad(d, gy, wb, None, dx))
for _ in range(10):
    run()
fn = lib.rdfc_debug_glds_tile_stamps
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(None, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * (512 * 10))()
assert fn(buf, 0) == 0
s = np.array(buf, dtype=np.uint64).reshape(512, 10).astype(np.int64)
s = s[s[:, 4] > 0]
t = s[:, :4] - s[:, 0].min()
span = t[:, 3].max()
life = t[:, 3] - t[:, 0]
print(f'{which} N{N} {H}x{W} {Ci}->{Co} kernel {lib.rdfc_last_kernel().decode()}: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(s)} sampled workgroups, launch span {span} ticks')
print(f'  per tile (mean ticks): load wait {np.mean(t[:, 1] - t[:, 0]):7.0f} | K loop {np.mean(t[:, 2] - t[:, 1]):7.0f} | epilogue {np.mean(t[:, 3] - t[:, 2]):7.0f} | life {life.mean():7.0f} (min {life.min()}, max {life.max()})')
print(f'  epilogue parts (first row group): barrier {np.mean(s[:, 8] - s[:, 2]):6.0f} | convert + stage {np.mean(s[:, 9] - s[:, 8]):6.0f} | read back + stores (+ further row groups) {np.mean(s[:, 3] - s[:, 9]):6.0f}')
real = (s[:, 6] - s[:, 5]) / 100.0                          # s_memrealtime: 100 MHz
clk = (t[:, 3] - t[:, 0]) / np.maximum(real, 1e-3) / 1e3
us = e0.elapsed_time(e1) * 1e3
print(f'  tile life by the 100 MHz clock: {real.mean():.2f} us (min {real.min():.2f}, max {real.max():.2f}); shader clock {np.median(clk):.2f} GHz')
print(f'  launch window by the 100 MHz clock: {(s[:, 6].max() - s[:, 5].min()) / 100.0:.1f} us (event time {us:.1f})')
order = np.argsort(t[:, 0])
print('  first tiles  (entry, landed, loop done, stored):', t[order[:3]].tolist())
print('  middle tiles (entry, landed, loop done, stored):', t[order[len(order) // 2:len(order) // 2 + 3]].tolist())

"""How much slack the host has: the train step with an artificial busy-wait of S microseconds added to EVERY kernel launch's host side
(the stream-handle accessor every launch goes through), S = 0, 2, 5, 10, 20.  While the step time does not move the host is not on the
critical path.  python tools/host_slack.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from rdfc_gan_amd import functional as Fh, hip
from rdfc_gan_amd.rdfc_gan import RDFCGAN

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
Fh.set_compute_dtype(torch.float16)
Fh.reserve_streams(dev, with_net=True)
m = RDFCGAN(dev, False, bench.model_cfg(False), is_train=True); m.train()
batches = [bench.synth_batch(8, 228, 304, 1234 + i, dev) for i in range(2)]
orig = hip._st
spin_us, calls = [0.0], [0]


def slow_st():
    calls[0] += 1
    if spin_us[0] > 0:
        t = time.perf_counter() + spin_us[0] * 1e-6
        while time.perf_counter() < t:
            pass
    return orig()


hip._st = slow_st
for i in range(6):
    m.set_input(batches[i % 2]); m.optimize_parameters()
for s in (0, 2, 5, 10, 20, 0):
    spin_us[0] = float(s)
    torch.cuda.synchronize(); calls[0] = 0
    t0 = time.perf_counter()
    for i in range(steps):
        m.set_input(batches[i % 2]); m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'+{s:2d} us per launch ({calls[0] // steps} launches per step through hip._st: +{s * calls[0] / steps * 1e-3:.1f} ms of host time per step): {dt * 1e3:.2f} ms per step, {8 / dt:.1f} pairs/s', flush=True)

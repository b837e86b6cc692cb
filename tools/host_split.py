"""Where the host time of a train step goes: wall time spent inside each autograd.Function's forward / backward (Python side,
both the calling thread and the autograd engine thread), plus everything else."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from rdfc_gan_amd import functional as Fh, rdfc_gan, hip
from rdfc_gan_amd.rdfc_gan import RDFCGAN
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name):
    f = getattr(cls, name)
    fn = f.__func__ if hasattr(f, '__func__') else f
    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[f'{cls.__name__}.{name}']; e[0] += time.perf_counter() - t; e[1] += 1
    setattr(cls, name, staticmethod(timed))
for mod in (Fh, rdfc_gan):
    for n in dir(mod):
        c = getattr(mod, n)
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
            wrap(c, 'forward'); wrap(c, 'backward')
from rdfc_gan_amd import nlspn
dev = torch.device('cuda', 0)
Fh.set_compute_dtype(torch.bfloat16)
m = RDFCGAN(dev, False, bench.model_cfg(False), is_train=True); m.train()
batches = [bench.synth_batch(8, 228, 304, 1234 + i, dev) for i in range(2)]
def step(i):
    m.set_input(batches[i % 2]); return m.optimize_parameters(sync_losses=False)
for i in range(4): step(i)
torch.cuda.synchronize(); acc.clear()
N = 8
host = wall = 0.0
for i in range(N):          # one step at a time into an IDLE queue: with several steps in flight the launch queue fills up and the
    t0 = time.perf_counter()  # enqueuing thread blocks inside whichever call comes next
    step(i)
    host += time.perf_counter() - t0
    torch.cuda.synchronize(); wall += time.perf_counter() - t0
print(f'per step: host enqueue {1e3*host/N:.2f} ms, wall {1e3*wall/N:.2f} ms')
tot = 0
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f'  {k:32s} {1e3*s/N:7.2f} ms/step  {n/N:6.1f} calls/step  {1e6*s/max(n,1):7.1f} us/call'); tot += s
print(f'  sum inside Function.forward/backward: {1e3*tot/N:.2f} ms/step; outside (module code, autograd engine, torch ops): {1e3*(host-tot)/N:.2f} ms')

"""cProfile of the host side of the eager train step (which Python code the launch-bound step spends its time in)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.rdfc_gan import RDFCGAN
dev = torch.device('cuda', 0)
Fh.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
m = RDFCGAN(dev, False, bench.model_cfg(False), is_train=True)
m.train()
batches = [bench.synth_batch(8, 228, 304, 1234 + i, dev) for i in range(2)]
def step(i):
    m.set_input(batches[i % 2]); return m.optimize_parameters(sync_losses=False)
for i in range(3): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(3): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats('cumulative'); ps.print_stats(60); print(s.getvalue()[:9000])

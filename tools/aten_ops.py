"""diagnostic: which stock (aten) device kernels a train step still launches, with the Python stack that issues them"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.rdfc_gan import RDFCGAN
dev = torch.device('cuda', 0)
Fh.set_compute_dtype(torch.bfloat16)
m = RDFCGAN(dev, False, bench.model_cfg('--nlspn' in sys.argv), is_train=True)
m.train()
batches = [bench.synth_batch(8, 228, 304, 50 + i, dev) for i in range(2)]
for it in range(3):
    m.set_input(batches[it % 2]); m.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.set_input(batches[1]); m.optimize_parameters()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    st = [s for s in e.stack if 'rdfc_gan_amd' in s or 'bench.py' in s or 'tools/' in s][:3]
    print(f'{e.key:28s} calls {e.count:4d} device {e.device_time_total:8.1f} us   ' + ' <- '.join(s.split('/')[-1] for s in st))

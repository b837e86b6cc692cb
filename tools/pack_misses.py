"""diagnostic: single-weight pack launches per train step (steady state: 0 -- everything goes through the batched launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.rdfc_gan import RDFCGAN
dev = torch.device('cuda', 0)
Fh.set_compute_dtype(torch.bfloat16)
m = RDFCGAN(dev, False, bench.model_cfg('--nlspn' in sys.argv), is_train=True)
m.train()
for it in range(4):
    before = Fh.pack_misses
    m.set_input(bench.synth_batch(2, 64, 80, 50 + it, dev))
    m.optimize_parameters()
    print(f'step {it}: single pack launches {Fh.pack_misses - before}, registry {len(Fh._pack_reg)}, tables {[(k[2], v[2], v[3]) for k, v in Fh._pack_tables.items()]}')

"""Two data-parallel ranks on ONE GPU over gloo (RCCL needs a GPU per rank): exercises the step's collectives, the side
streams and FlatAdam's bucket exchange end to end; after two steps the replicas must hold identical parameters.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/dp_smoke.py"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.rdfc_gan import RDFCGAN

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
Fh.set_compute_dtype(torch.bfloat16)
torch.manual_seed(100 + rank)           # different initial weights per rank: the broadcast must equalise them
random.seed(rank); np.random.seed(rank)
full = os.environ.get('DP_SMOKE_FULL') == '1'          # all six networks: label net + G_normal (stand-in encoder), three more reducers
kw = {}
if full:
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from cases import StandInEncoder
    from rdfc_gan_amd.normal_net import NNET
    kw = dict(use_label_net=True, normal_net=NNET(64, 80, encoder=StandInEncoder()), out_height=64, out_width=80)
m = RDFCGAN(dev, True, bench.model_cfg(False), is_train=True, **kw)
m.train()
for it in range(2):
    m.set_input(bench.synth_batch(2, 64, 80, 50 + 10 * rank + it, dev))
    losses = m.optimize_parameters()
    assert all(np.isfinite(v) for v in losses.values()), losses
torch.cuda.synchronize()
flat = torch.cat([o.flat for o in m.optimizers.values()]).float()
mine = flat.cpu()
other = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(other, mine)
diff = max((o - other[0]).abs().max().item() for o in other)
if rank == 0:
    print(f'dp_smoke: optimizers {sorted(m.optimizers)}; world {world}, max parameter difference between replicas after 2 steps = {diff:.3e}; losses {losses}')
assert diff == 0.0, diff
dist.destroy_process_group()

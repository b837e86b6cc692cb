"""Run many train steps and watch allocator statistics (leak / fragmentation check for the multi-stream step) and the loss trajectory.
SOAK_DTYPE=f16 | bf16 | f32 (f32 = the parity mode: the trajectory the 16-bit modes are read against, same seeds and batches);
SOAK_EVERY=n prints every n-th step (a multiple of 20).  python tools/soak.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.rdfc_gan import RDFCGAN
dev = torch.device('cuda', 0)
Fh.set_compute_dtype(dict(bf16=torch.bfloat16, f16=torch.float16, f32=torch.float32)[os.environ.get('SOAK_DTYPE', 'f16')])
torch.manual_seed(0)
import random, numpy as np
random.seed(1234); np.random.seed(1234)            # (the pepper-noise masks: the same draws in every mode)
m = RDFCGAN(dev, False, bench.model_cfg(False), is_train=True); m.train()
batches = [bench.synth_batch(8, 228, 304, 1234 + i, dev) for i in range(2)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(n):
    m.set_input(batches[i % 2]); ls = m.optimize_parameters(sync_losses=(i % 20 == 19))
    if i % 20 == 19 and (i + 1) % int(os.environ.get('SOAK_EVERY', 100 if n > 400 else 20)) == 0:
        torch.cuda.synchronize()
        print(f'step {i+1}: allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB, '
              f'peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, loss_G {ls["loss_G"]:.3f} (' + ', '.join(f'{k[5:]} {ls[k]:.4f}' for k in ('loss_G_A2B', 'loss_G_B2A', 'loss_cycle_A', 'loss_cycle_B', 'loss_A2B_L1', 'loss_D_A', 'loss_D_B') if k in ls) + ')'
              + (f', loss scale {m.scaler.scale():.0f}, skipped {m.scaler.skipped()}' if m.scaler is not None else ''))

"""NLSPN-shaped modulated deformable conv backward: LDS-window kernel vs the generic one (RDFC_MDCONV_GENERIC=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip
B, H, W = 8, 228, 304
torch.manual_seed(0)
x = torch.randn(B, 1, H, W, device='cuda'); off = torch.randn(B, 18, H, W, device='cuda') * float(sys.argv[1] if len(sys.argv) > 1 else 0.5)
msk = torch.rand(B, 9, H, W, device='cuda'); w = torch.ones(1, 1, 3, 3, device='cuda'); go = torch.randn(B, 1, H, W, device='cuda')
def run(): return hip.mdconv_backward(x, off, msk, w, go, (1, 1), (1, 1), (1, 1), 1, 1, need=(True, True, True, False, False))
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); print(f'mdconv backward 8x1x228x304 k3: {(time.perf_counter() - t) / 20 * 1e6:.1f} us per call (incl. memset)')

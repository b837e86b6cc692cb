"""Deterministic, construction-order-independent weights for parity tests.

Every state_dict entry is drawn from its own generator seeded by a hash of
(seed, key), so the reference (imported in the build container by
make_golden.py), the oracle and the HIP modules can all be loaded with exactly
the same values without committing 100+ MB of weights.
"""
import zlib

import torch


def _gen(seed, key):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def det_state_dict(template, seed=0):
    """template: a state_dict (only key / shape / dtype are used)."""
    out = {}
    for key, t in template.items():
        g = _gen(seed, key)
        leaf = key.rsplit('.', 1)[-1]
        shp = tuple(t.shape)
        if leaf == 'num_batches_tracked':
            v = torch.zeros(shp, dtype=t.dtype)
        elif leaf == 'running_mean':
            v = 0.05 * torch.randn(shp, generator=g)
        elif leaf == 'running_var':
            v = 0.8 + 0.4 * torch.rand(shp, generator=g)
        elif leaf == 'weight_orig':                      # EqualLinear (N(0,1) in the reference)
            v = torch.randn(shp, generator=g)
        elif key.endswith('style.linear.bias'):          # gamma ~ 1, beta ~ 0
            v = 0.05 * torch.randn(shp, generator=g)
            v[: shp[0] // 2] += 1.0
        elif leaf == 'aff_scale_const':
            v = torch.full(shp, 4.0) + 0.1 * torch.randn(shp, generator=g)
        elif leaf in ('w', 'w_conf'):                    # NLSPN frozen gather kernels
            v = torch.ones(shp)
        elif leaf == 'b':
            v = torch.zeros(shp)
        elif 'conv_offset_aff' in key:                   # zero-init upstream; small non-zero here to exercise offsets
            v = (0.03 if leaf == 'weight' else 0.1) * torch.randn(shp, generator=g)   # (contractive propagation regime)
        elif leaf == 'weight' and len(shp) == 1:         # BN gamma or PReLU slope
            v = (0.25 + 0.02 * torch.randn(shp, generator=g)) if shp == (1,) else (1.0 + 0.1 * torch.randn(shp, generator=g))
        elif leaf == 'weight':                           # conv / convT / linear
            fan = 1
            for d in shp[1:]:
                fan *= d
            v = torch.randn(shp, generator=g) * (1.0 / max(fan, 1)) ** 0.5
        elif leaf == 'bias':
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            raise KeyError(f'no rule for state_dict key {key}')
        out[key] = v.to(t.dtype)
    return out


def det_tensor(seed, key, shape, kind='randn'):
    g = _gen(seed, key)
    return torch.randn(shape, generator=g) if kind == 'randn' else torch.rand(shape, generator=g)

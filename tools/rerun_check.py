"""GPU diagnostic: is G_A2B's eval forward reproducible run to run, and if not, which launch is the first to differ?

The forward runs N times on the same weights and inputs with a device synchronisation after every call; run 0's tensors are kept
(inputs and results of every functional call and of every hip.conv_forward / hip.norm_apply launch) and every later run is compared
with them element by element.  `python tools/rerun_check.py [f16|bf16|f32] [H W] [runs]`.  Environment: FINAL=1 -- no instrumentation at all,
only the five results of every run are compared (the fast yes / no); INNER=0 -- wrap the functional calls only, not the launches inside
them; LIMIT=n -- differing calls reported per run (default 3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch

from cases import gen_inputs
from detweights import det_state_dict
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd import hip
from rdfc_gan_amd.generator import RDFGenerator

ref, pos, reported = [], [0], [0]
LIMIT = int(os.environ.get('LIMIT', '3'))


def tensors(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in tensors(x)]
    return []


def wrap(mod, name, out_args=None):
    f = getattr(mod, name)

    def g(*a, **kw):
        out = f(*a, **kw)
        torch.cuda.synchronize()
        outs = [a[i] for i in out_args] if out_args else [t for t in tensors(out) if t.is_floating_point()]
        ins = [t for i, t in enumerate(a) if torch.is_tensor(t) and t.is_floating_point() and not (out_args and i in out_args)]
        i = pos[0]
        pos[0] += 1
        if i >= len(ref):
            ref.append((name, [t.clone() for t in ins], [t.clone() for t in outs]))
            return out
        rname, rin, rout = ref[i]
        assert rname == name
        same_in = all(p.shape == q.shape and torch.equal(p, q) for p, q in zip(rin, ins))
        for k, (p, q) in enumerate(zip(rout, outs)):
            if not torch.equal(p, q) and reported[0] < LIMIT:
                reported[0] += 1
                dd = (p.float() - q.float()).abs()
                idx = (dd > 0).nonzero()
                msg = (f'   launch {i} {name} out[{k}] {tuple(q.shape)} stream {torch.cuda.current_stream().cuda_stream:#x} inputs identical: {same_in}; '
                       f'{idx.shape[0]} elements differ, max {float(dd.max()):.3e}; ch {int(idx[:, 3].min())}..{int(idx[:, 3].max())}')
                if same_in:
                    tiles = sorted({(int(y) // 16, int(x) // 16) for y, x in idx[:, 1:3].tolist()})
                    chans = sorted({int(c) for c in idx[:, 3].tolist()})
                    msg += f'\n      16x16 tiles (ty, tx) {tiles[:40]}{" ..." if len(tiles) > 40 else ""} ({len(tiles)} tiles)\n      channels {chans}'
                    t0 = tiles[0]
                    sub = idx[(idx[:, 1] // 16 == t0[0]) & (idx[:, 2] // 16 == t0[1])]
                    msg += f'\n      in tile {t0}: rows {sorted({int(v) % 16 for v in sub[:, 1].tolist()})} cols {sorted({int(v) % 16 for v in sub[:, 2].tolist()})}'
                print(msg, flush=True)
        return out
    setattr(mod, name, g)


def main():
    dt = dict(f16=torch.float16, bf16=torch.bfloat16, f32=torch.float32)[sys.argv[1] if len(sys.argv) > 1 else 'f16']
    H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    Fh.set_compute_dtype(dt)
    Fh.invalidate_packed_weights()
    net = RDFGenerator(pretrained_on_imagenet=False)
    net.load_state_dict(det_state_dict(net.state_dict(), 0))
    net.to('cuda').eval()
    inputs = [t.cuda() for t in gen_inputs('cfg5', 1, H, W)]
    if os.environ.get('FINAL', '0') == '1':        # no instrumentation at all: only the five results of every run are compared
        first = None
        for r in range(runs):
            with torch.no_grad():
                out = {k: v.float().clone() for k, v in net(*inputs).items()}
            torch.cuda.synchronize()
            if first is None:
                first = out
            else:
                print(f'run {r}: ' + ' '.join(f'{k}={float((out[k] - first[k]).abs().max()):.1e}' for k in out), flush=True)
        return
    for n in ('conv_block', 'wadain', 'cat', 'crop_cat', 'to_pm', 'fusion_head'):
        wrap(Fh, n)
    if os.environ.get('INNER', '1') == '1':
        wrap(hip, 'conv_forward', (2, 4))          # (d, x, wf, bias, y): the packed weights count as a result too
        wrap(hip, 'norm_apply', (6,))
    for r in range(runs):
        pos[0], reported[0] = 0, 0
        with torch.no_grad():
            net(*inputs)
        torch.cuda.synchronize()
        print(f'run {r}: {pos[0]} launches, {"identical" if reported[0] == 0 else "DIFFERENT"}', flush=True)


if __name__ == '__main__':
    main()

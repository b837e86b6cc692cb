"""Two data-parallel ranks on ONE GPU over gloo (RCCL needs a GPU per rank): exercises the step's collectives, the side
streams and FlatAdam's bucket exchange end to end; after two steps the replicas must hold identical parameters.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/dp_smoke.py

DP_SMOKE_BACKEND=nccl with ONE rank (RDFC_DP_FORCE=1 is set here): the same step with every collective going through RCCL on the
real device (broadcast, bucketed all-reduces on the communication stream, packed loss all-reduces); its parameters after two steps
must equal those of the same step run without a process group, bit for bit."""
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
backend = os.environ.get('DP_SMOKE_BACKEND', 'gloo')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
if backend == 'nccl':
    assert world == 1, 'RCCL wants one GPU per rank; this box has one'
    os.environ['RDFC_DP_FORCE'] = '1'
    dist.init_process_group('nccl', device_id=dev)
else:
    dist.init_process_group('gloo')
Fh.set_compute_dtype(dict(bf16=torch.bfloat16, f16=torch.float16)[os.environ.get('DP_SMOKE_DTYPE', 'bf16')])
graph = os.environ.get('DP_SMOKE_GRAPH') == '1'       # the step captured into HIP graphs after one eager step, then replayed
nsteps = 5 if graph else 2
torch.manual_seed(100 + rank)           # different initial weights per rank: the broadcast must equalise them
random.seed(rank); np.random.seed(rank)
full = os.environ.get('DP_SMOKE_FULL') == '1'          # all six networks: label net + G_normal (stand-in encoder), three more reducers
kw = {}
if full:
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from cases import StandInEncoder
    from rdfc_gan_amd.normal_net import NNET
    kw = dict(use_label_net=True, normal_net=NNET(64, 80, encoder=StandInEncoder()), out_height=64, out_width=80)


# DP_SMOKE_DELAY_MAIN=1 (round-4 advice): G_A2B's decoder-side bucket is started by whichever branch's backward marker fires last --
# from the depth branch's stream -- while the RGB branch's last decoder-side node (fuse_layer1's W-AdaIN) hands its weight gradients
# to autograd, which accumulates them on the MAIN stream after the event the depth branch waits for.  A gradient hook on the fusion
# layers' parameters puts a long spin kernel exactly there (between that event and the accumulation): without the event that the RGB
# marker records on its own stream the collective reads the range before the add has landed, the add then lands on the reduced
# values, and the replicas drift apart (DP_SMOKE_NO_MARK_EVENTS=1 drops the events: the run must then FAIL the comparison below).
def delay_fusion_grads(m):
    if os.environ.get('DP_SMOKE_DELAY_MAIN') != '1':
        return
    def slow(g):
        torch.cuda._sleep(int(4e7))          # ~20 ms
        return g
    n = 0
    for name, p_ in m.G_A2B.named_parameters():
        if 'fuse_layer' in name:
            p_.register_hook(slow)
            n += 1
    if rank == 0 and os.environ.get('DP_SMOKE_TRACE') == '1':          # which stream each marker fires on, in host order
        print(f'dp_smoke: {n} fusion-layer parameters delayed; main stream {torch.cuda.current_stream(dev).cuda_stream:#x}')
        orig = m._on_G_mark

        def traced(name):
            print(f'dp_smoke: marker {name} fires on stream {torch.cuda.current_stream(dev).cuda_stream:#x}', flush=True)
            return orig(name)
        m._on_G_mark = traced


if os.environ.get('DP_SMOKE_NO_MARK_EVENTS') == '1':
    from rdfc_gan_amd import optim as _optim
    _orig_ar = _optim.FlatAdam.all_reduce_bucket_async
    _optim.FlatAdam.all_reduce_bucket_async = lambda self, i, helpers=True, after=(): _orig_ar(self, i, helpers)


def two_steps(distributed):
    torch.manual_seed(100 + rank)
    random.seed(rank); np.random.seed(rank)
    m = RDFCGAN(dev, distributed, bench.model_cfg(False), is_train=True, **kw)
    m.train()
    delay_fusion_grads(m)
    trace = []
    for it in range(nsteps):
        if graph and it == 1:
            m.enable_graph(True)
        m.set_input(bench.synth_batch(2, 64, 80, 50 + 10 * rank + it, dev))
        losses = m.optimize_parameters()
        assert all(np.isfinite(v) for v in losses.values()), losses
        trace.append(losses)
    torch.cuda.synchronize()
    if graph:
        assert m._graph_mode and m._graphs is not None, 'capture fell back to eager launches'
        m.enable_graph(False)
    return m, trace


m, trace = two_steps(True)
losses = trace[-1]
flat = torch.cat([o.flat for o in m.optimizers.values()]).float()
# the issue order of the last eager step's gradient buckets: G = [G_B2A, G_A2B decoder side, deep encoder stages, the rest] -- the first
# three are started from INSIDE the backward (markers), i.e. they are in flight before it ends; the discriminators' buckets likewise
if not graph and not full:
    oG = m.optimizers['G']
    assert len(oG._buckets) == 4 and oG.started_order == [0, 1, 2, 3], (len(oG._buckets), oG.started_order)
    assert m.__dict__['_G_marks'] == dict(dec=2, deep=2), m.__dict__['_G_marks']
    sizes = [sum(b - a for a, b in oG.bucket_ranges(i)) * 4e-6 for i in range(4)]
    assert sizes[2] > 2 * sizes[1] > 0 and sizes[2] > 2 * sizes[3] > 0, sizes           # the deep stages are the bulk (86 of 168 MB)
    if rank == 0:
        print(f'dp_smoke: G buckets in issue order (MB): {[round(x, 1) for x in sizes]}')
if backend == 'nccl':
    from rdfc_gan_amd.optim import dp_active
    assert dp_active() and dist.get_backend() == 'nccl'
    started = {n: sorted(o._started) for n, o in m.optimizers.items()}
    assert all(started[n] == list(range(len(o._buckets))) for n, o in m.optimizers.items()), started     # every bucket went out
    os.environ['RDFC_DP_FORCE'] = '0'
    m2, trace2 = two_steps(False)
    flat2 = torch.cat([o.flat for o in m2.optimizers.values()]).float()
    # Step 1's losses depend on the initial weights only (the one-rank broadcast must leave them alone): equal bit for bit.
    # After the updates the two runs differ by what the weight gradients' split-K float atomics differ between ANY two runs
    # (an Adam update of a noise-level gradient can flip: 2 lr per step), so the parameters are compared against that bound.
    d, mean = (flat - flat2).abs().max().item(), (flat - flat2).abs().mean().item()
    first = max(abs(trace[0][k] - trace2[0][k]) for k in trace[0])
    last = max(abs(trace[1][k] - trace2[1][k]) / max(1.0, abs(trace2[1][k])) for k in trace[1])
    print(f'dp_smoke: RCCL one-rank step, buckets {started}; step-1 loss difference to the step without a process group = {first:.3e}; '
          f'parameters after 2 steps: max difference {d:.3e}, mean {mean:.3e}; step-2 losses relative {last:.3e}')
    assert first == 0.0 and d <= 1e-2 and mean <= 1e-3 and last <= 2e-2, (first, d, mean, last)
    dist.destroy_process_group()
    sys.exit(0)
mine = flat.cpu()
other = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(other, mine)
diff = max((o - other[0]).abs().max().item() for o in other)
if rank == 0:
    print(f'dp_smoke: optimizers {sorted(m.optimizers)}; world {world}, max parameter difference between replicas after {nsteps} steps = {diff:.3e}; losses {losses}')
assert diff == 0.0, diff
dist.destroy_process_group()

#!/usr/bin/env python
"""bench.py -- RGB-D pairs/s per train step (G+D) of the RDFC-GAN hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = RDFCGAN.optimize_parameters on one synthetic batch (8 pairs per GPU at 228x304,
BASELINE.json configs[1]): G_A2B + G_B2A + D_A + D_B forward, backward, Adam.  Weak scaling: every
rank draws its own shard; gradients are all-reduced with RCCL on a side stream.
Prints ONE JSON line (rank 0) with the metric, a roofline block for the dominant kernel family
(per-launch HIP events inside the timed region) and a CPU baseline (the oracle timed on the host).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

# forward GFLOP per pair (2*MAC over conv / conv-transpose / linear) at 228x304
GFLOP_FWD = dict(G_A2B=209.64, G_B2A=73.28, D_A=6.49, D_B=6.56, G_label=72.49)   # G_label: train mode, both heads
PEAK_TFLOPS = dict(bf16=2500.0, f16=2500.0, f32=157.3)        # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                                         # HBM3E spec (6 290 GB/s measured by a float4 copy), MI355X_MICROARCH.md


def _log(msg):
    """progress on stderr (rank 0): the default run takes several minutes -- headline, three extra legs, a CPU baseline of ~6 minutes"""
    if int(os.environ.get('RANK', 0)) == 0:
        print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def pmc_table(suffix=''):
    """the newest committed rocprofv3 --pmc summary of a workload: profiles/rNN_pmc_traffic<suffix>.json (tools/pmc_bench.sh; suffix '' = the
    headline command, '_bf16', '_nlspn', '_480x640' = the extra legs).  -> (table without _meta, _meta, file name) or (None, None, None)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_pmc_traffic{suffix}.json')):
        m = re.match(r'r(\d\d)_pmc_traffic' + re.escape(suffix) + r'\.json$', os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None, None, None
    tab = json.load(open(best[1]))
    meta = tab.pop('_meta', {})
    return tab, meta, os.path.relpath(best[1], ROOT)


def pmc_for_kernel(roof, top, suffix, loaded_hash):
    """fill roofline['traffic'] / ['mfma_busy'] of the dominant kernel `top` from the workload's PMC summary; counters taken from other
    kernel sources than the loaded library's are refused.  -> the summary's _meta (or None)"""
    tab, meta, fname = pmc_table(suffix)
    if tab is None:
        return None
    if meta.get('csrc_sha16') != loaded_hash:
        roof['traffic_note'] = (f"{fname} was measured on kernel sources {meta.get('csrc_sha16')}, the loaded library was built from "
                                f'{loaded_hash}: stale counters refused (re-run tools/pmc_bench.sh)')
        return None
    for k, v in tab.items():
        if top + '(' in k:
            roof['traffic'] = round(v['fetch_bytes'] + v['write_bytes'])
            roof['traffic_note'] = (f"bytes per launch (read {v['fetch_bytes'] / 1e6:.1f} MB + write {v['write_bytes'] / 1e6:.1f} MB), "
                                    f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh over this workload ({fname}), kernel sources {meta.get('csrc_sha16')}")
            if v.get('mfma_busy_cycles'):
                # third --pmc pass of the same script: cycles the matrix pipes were busy, summed over the chip's 1 024 SIMDs, per launch,
                # against the SIMD-cycles of this run's average launch at the 2.4 GHz peak clock (an upper bound for the clock, so a
                # lower bound for the busy fraction)
                simd_cycles = roof['avg_launch_us'] * 2400.0 * 1024
                roof['mfma_busy'] = dict(cycles_per_launch=round(v['mfma_busy_cycles']), mfma_insts_per_launch=round(v['mfma_insts']),
                                         frac_of_simd_cycles=round(v['mfma_busy_cycles'] / simd_cycles, 4),
                                         note=f'SQ_VALU_MFMA_BUSY_CYCLES per launch ({fname}) / (avg_launch_us x 2.4 GHz x 1024 SIMDs)')
    return meta


def step_gflop_per_pair(exact_schedule, scale=1.0, label_net=False):
    """G nets: fwd x3 (x2 when the dead forwards are elided) + bwd x2 (bwd = 2x fwd);
    D nets: fwd x3 + input-gradient-only pass x1 + full bwd x2; G_label (--label-net): fwd x2 + bwd x2."""
    nf = 3 if exact_schedule else 2
    g = (GFLOP_FWD['G_A2B'] + GFLOP_FWD['G_B2A']) * (nf + 2 * 2)
    d = (GFLOP_FWD['D_A'] + GFLOP_FWD['D_B']) * (3 + 1 + 2 * 2)
    lab = GFLOP_FWD['G_label'] * (2 + 2 * 2) if label_net else 0.0
    return (g + d + lab) * scale


def synth_batch(B, H, W, seed, device):
    """NYUv2-shaped synthetic batch (same keys / ranges as the upstream loader)."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    m = (2.84 + 0.99 * torch.randn(B, 1, H, W, generator=g))
    m = F.avg_pool2d(F.pad(m, (4, 4, 4, 4), mode='replicate'), 9, 1).clamp(0.713, 9.995)
    gt = (m - 5.0) / 5.0
    raw = gt.clone()
    for b in range(B):                      # ~30 % of the pixels in random rectangles -> 0 m -> -1.0
        for _ in range(6):
            y0, x0 = int(torch.randint(0, H - 8, (1,), generator=g)), int(torch.randint(0, W - 8, (1,), generator=g))
            hh, ww = int(torch.randint(8, H // 3, (1,), generator=g)), int(torch.randint(8, W // 3, (1,), generator=g))
            raw[b, :, y0:y0 + hh, x0:x0 + ww] = -1.0
    normal = F.normalize(torch.randn(B, 3, H, W, generator=g), dim=1)
    data = dict(rgb=rgb, gt_depth=gt, raw_depth=raw, depth_masks=m > 1e-4, gt_normal=normal,
                labels=torch.randint(0, 14, (B, H // 12 + 1, W // 12 + 1), generator=g).repeat_interleave(12, 1).repeat_interleave(12, 2)[:, :H, :W]
                .contiguous(), normal_masks=torch.ones(B, 1, H, W))
    return {k: v.to(device) for k, v in data.items()}


def model_cfg(nlspn):
    from cases import NLSPN_CFG
    ga2b = dict(type='RDFGenerator', pretrained_on_imagenet=False)
    if nlspn:
        ga2b.update(use_nlspn_refine=True, nlspn_configs=NLSPN_CFG)
    return dict(local_rank=int(os.environ.get('LOCAL_RANK', 0)), lr=2e-3, beta1=0.5, beta2=0.999, gan_loss_type='lsgan',
                lambda_A=100.0, lambda_B=100.0, lambda_L1=100.0,
                model=dict(G_A2B=ga2b,
                           G_B2A=dict(type='ResnetGenerator', input_channels=1, output_channels=3, ngf=64, norm_layer='IN2d'),
                           D_A=dict(type='PatchGANDiscriminator', in_channels=1, norm_cfg=dict(type='BN2d'), activation='LeakyReLU'),
                           D_B=dict(type='PatchGANDiscriminator', in_channels=3, norm_cfg=dict(type='BN2d'), activation='LeakyReLU')))


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def cpu_baseline(B, H, W, exact_schedule=True, budget_s=150.0, steps=0):
    """The oracle's train step (plain torch.nn on the host cores; SURVEY 8d / BASELINE.md section 4): the SAME batch size and
    resolution as the GPU run, upstream's literal schedule (dead forwards included), fp32.  Bounded sample: one warm-up step at
    batch 1 (thread pools, allocator, lazy initialisation -- a batch-8 step takes ~2 minutes on the GPU box's 128 threads), then
    timed steps at the full batch for as long as `budget_s` allows (at least one; three when they fit), median.  steps > 0
    (--cpu-steps): exactly that many timed steps whatever they cost -- BASELINE.md section 4's protocol is 3 (about 6 minutes of
    host time on the GPU box, which is why the default run keeps to the budget and says so in the line)."""
    from oracle import nets as onets
    from oracle.step import HotPathStep
    torch.manual_seed(0)
    nets = (onets.RDFGenerator(), onets.ResnetGenerator(1, 3, 64, 'IN2d'), onets.PatchGANDiscriminator(1),
            onets.PatchGANDiscriminator(3))
    for n in nets:
        n.train()
    st = HotPathStep(*nets)

    def one(batch):
        data = synth_batch(batch, H, W, 999, 'cpu')
        st.set_input(data)
        nrm = data['gt_normal']
        t0 = time.perf_counter()
        st.step(lambda tag: nrm)
        return time.perf_counter() - t0

    _log(f'cpu_baseline: 1 warm-up step at batch 1 + {steps if steps > 0 else "up to 3"} timed step(s) at batch {B} on {torch.get_num_threads()} threads '
         f'(~2 minutes each on a 128-thread GPU-box host; --cpu-steps 0 keeps to a {budget_s:.0f} s budget, --no-cpu-baseline skips it) ...')
    warm = one(1)
    times = [one(B)]
    _log(f'cpu_baseline: step 1 took {times[0]:.1f} s')
    while len(times) < steps if steps > 0 else (len(times) < 3 and sum(times) + times[-1] <= budget_s):
        times.append(one(B))
        _log(f'cpu_baseline: step {len(times)} took {times[-1]:.1f} s')
    times.sort()
    med = times[len(times) // 2]
    return dict(value=round(B / med, 5), unit='pairs/s', cores=torch.get_num_threads(), kind='port',
                steps=len(times), warmup_steps=1, warmup_batch=1, budget_s=None if steps > 0 else budget_s,
                protocol='BASELINE.md section 4: 1 warm-up + 3 timed steps' + ('' if len(times) >= 3 else f'; cut to {len(times)} timed step(s) by the {budget_s:.0f} s budget of the default run (--cpu-steps 3 runs it in full)'),
                sample=f'{len(times)} timed train step(s) at batch {B} (median {med:.1f} s) after one warm-up step at batch 1 ({warm:.1f} s); '
                       f'upstream schedule incl. dead forwards, {H}x{W}, fp32 torch CPU oracle, torch threads {torch.get_num_threads()} on '
                       f'{os.cpu_count()} logical CPUs ({_cpu_model()})')


def rmse_deviation(model, H, W, dev):
    """North-star accuracy figure for the benchmarked storage type: |dense-depth RMSE (metres) of the eval path (forward_test ->
    RDFGANMetric) - the same on the SAME weights and batch in the f32 parity mode|, max over 3 synthetic batches x 2 images.  The
    f32 mode is what tests/test_modules_gpu.py::test_eval_path_end_to_end_* pins to the CPU oracle at 1e-4 m; nothing of oracle/
    runs here."""
    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd.metrics import RDFGANMetric
    fast, worst = Fh.compute_dtype(), 0.0
    if fast == torch.float32:
        return 0.0
    model.eval()
    try:
        for k in range(3):
            b = synth_batch(2, H, W, 4242 + k, dev)
            rm = {}
            for dt in (fast, torch.float32):
                Fh.set_compute_dtype(dt)
                Fh.invalidate_packed_weights()
                with torch.no_grad():
                    out = model.forward_test(rgb=b['rgb'], raw_depth=b['raw_depth'], gt_normal=b['gt_normal'])
                rm[dt] = RDFGANMetric().evaluate_normalised(b['gt_depth'], out['pred_depth'], None).double().cpu().numpy()[:, 0]
            worst = max(worst, float(np.abs(rm[fast] - rm[torch.float32]).max()))
    finally:
        Fh.set_compute_dtype(fast)
        Fh.invalidate_packed_weights()
        model.train()
    return worst


def _count_gpus():
    """GPUs of this node WITHOUT touching HIP (the parent must not initialise the runtime before it spawns the ranks): the KFD
    topology lists every agent; GPU nodes are the ones with SIMDs.  Falls back to torch's count (NVML-style, no context) when
    the topology is not readable."""
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(line.split()[:2] for line in open(os.path.join(base, node, 'properties')) if len(line.split()) >= 2)
            if int(props.get('simd_count', 0)) > 0:
                n += 1
        vis = os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES'))
        if vis is not None and vis.strip() != '':
            n = min(n, len([v for v in vis.split(',') if v.strip() != '']))
        return n
    except (OSError, ValueError):
        return torch.cuda.device_count()


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run job (one process per GPU,
    RCCL) before this process has touched the GPU -- a process that has initialised HIP must not exec another program on this
    pool --, relay the ranks' output (rank 0 prints the JSON line last) and return the job's exit code."""
    import socket
    import subprocess
    have = _count_gpus()
    if have < n:
        print(f'bench.py: --gpus {n} but this node shows {have} GPU(s): refusing to report a smaller job under that label', file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def comm_in_step(model, one_step, steps, dev):
    """what the gradient exchange costs INSIDE the step (untimed extra steps, every rank): (1) four traced steps -- per bucket, HIP events for
    "dependencies met" on the issue stream and around the consuming stream's wait (rdfc_gan_amd/optim.py COMM_TRACE): when in the step the
    bucket could start, when its consumer (the optimizer's Adam kernel) asked for it and how long that stream then stalled = the part of the
    exchange that nothing overlapped; (2) the same K steps with the collectives switched off (COLLECTIVES_OFF: same control flow and streams,
    gradients stay local): the difference to the timed figure is the whole cost of the exchange, overlap lost to RCCL's kernels included."""
    from rdfc_gan_amd import optim
    names = {id(o): n for n, o in model.optimizers.items()}
    rows = {}
    for i in range(4):
        optim.COMM_TRACE = []
        torch.cuda.synchronize()
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        one_step(i)
        torch.cuda.synchronize()
        for order, t in enumerate(optim.COMM_TRACE):
            r = rows.setdefault((names[id(t['optimizer'])], t['bucket']), dict(order=order, ready=[], asked=[], stall=[]))
            if t['ready'] is not None:
                r['ready'].append(start.elapsed_time(t['ready']))
            r['asked'].append(start.elapsed_time(t['wait_begin']))
            r['stall'].append(t['wait_begin'].elapsed_time(t['wait_end']))
    optim.COMM_TRACE = None
    med = lambda v: round(sorted(v)[len(v) // 2], 3) if v else None
    in_step = [dict(optimizer=k[0], bucket=k[1], wait_order=r['order'], ready_at_ms=med(r['ready']), consumer_asks_at_ms=med(r['asked']),
                    exposed_wait_ms=med(r['stall'])) for k, r in sorted(rows.items(), key=lambda kv: kv[1]['order'])]
    # (the flag is a module global: whatever happens inside, later collectives must not stay switched off; and the steps below apply every
    #  rank's LOCAL gradients, so weights, Adam moments and the loss-scaler state drift apart -- rank 0's state is broadcast again afterwards)
    optim.COLLECTIVES_OFF = True
    try:
        one_step(0)
        torch.cuda.synchronize()
        if dist.get_world_size() > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            one_step(i)
        torch.cuda.synchronize()
        off = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    finally:
        optim.COLLECTIVES_OFF = False
    optim.resync_replicas(model.models.values(), model.optimizers.values())
    dist.all_reduce(off, op=dist.ReduceOp.MAX)
    return dict(in_step=in_step, exposed_wait_ms_per_step=round(sum(r['exposed_wait_ms'] or 0.0 for r in in_step), 3),
                ms_per_step_collectives_off=round(1e3 * float(off) / steps, 3),
                note='in_step: medians over 4 traced steps, times from the step\'s first launch; exposed_wait = how long the consuming stream stalled for '
                     'the bucket; ms_per_step_collectives_off: the same step with every collective skipped (max over ranks) -- compare with ms_per_step; '
                     'the replicas are re-synchronised from rank 0 after that leg (parameters, buffers, Adam moments, step counts, loss scale)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=8, help='untimed steps (a cold first process on a fresh box needs a few: page cache, clocks)')
    ap.add_argument('--batch', type=int, default=8, help='pairs per GPU')
    ap.add_argument('--height', type=int, default=228)
    ap.add_argument('--width', type=int, default=304)
    ap.add_argument('--dtype', choices=('bf16', 'f16', 'f32'), default='f16',
                    help='storage / MFMA operand type of the conv operands.  f16 (default) is NOT the type BASELINE configs[1] words (bf16): same width, '
                         'same MFMA rate, same kernels and speed, 11 instead of 8 significant bits in the stored WEIGHTS -- the CPU simulation '
                         'tools/lowp_sim.py puts the eval-path RMSE shift of bf16 weights at up to ~1.4e-3 m by weight draw against <= 3.3e-4 m for '
                         'f16 (profiles/r03_lowp_sim.txt; `rmse_dev_m` of one bench run is measured on near-random weights and cannot rank the two); '
                         'the default run appends the bf16 figure of the same workload as `as_worded_bf16`.  bf16: configs[1] as worded')
    ap.add_argument('--nlspn', action='store_true', help='BASELINE configs[2]: NLSPN refinement on')
    ap.add_argument('--exact-schedule', action='store_true', help="run upstream's two dead forwards too")
    ap.add_argument('--label-net', action='store_true', help='SURVEY 8f rank 1 (not the headline workload): G_label (PSPNet) twice '
                    'per step + label cross entropy + normal-consistency terms + its Adam')
    ap.add_argument('--full-objective', action='store_true', help="--label-net plus G_normal (decoder on HIP behind the tests' stock-PyTorch "
                    'stand-in encoder): every term of upstream loss_G, six networks, four optimizers (not the headline workload)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed steps of the CPU baseline (default 3 = BASELINE.md section 4: 1 warm-up + 3 timed steps, '
                    '~6 minutes of host time on the GPU box); 0 = as many as fit a 150 s budget, at least one -- the line says which')
    ap.add_argument('--no-as-worded', action='store_true', help='skip the extra legs (untimed by the contract) that the default f16 run appends: the same '
                    'workload in bf16 (BASELINE configs[1] as worded: `as_worded_bf16`, with its own roofline block) and `configs` = configs[2] (NLSPN) '
                    "and configs[4]'s per-GPU shape (480x640, batch 4, f16)")
    ap.add_argument('--rccl-self-test', action='store_true', help='N = 1 only: initialise a one-rank RCCL process group and send the step '
                    'through every collective of the data-parallel path (bucketed all-reduces, loss all-reduces); adds the comm block')
    ap.add_argument('--no-sync-losses', action='store_true', help='do not read the losses back every step (the host then runs ahead '
                    'of the GPU; NOT the reference API behaviour -- the default run reports this variant as an extra key)')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--graph', action='store_true', help='capture the step into HIP graphs and replay them (single stream; on ROCm 7.2 the '
                    'replay costs about as much per node as an eager launch, so the multi-stream eager step is faster)')
    ap.add_argument('--event-every', type=int, default=20, help='per-kernel HIP events on every n-th timed step, the first one included (eager mode; such a step takes ~8 % longer: ~1 000 event pairs and the normalisation layers as one call per kernel)')
    ap.add_argument('--n1-value', type=float, default=None, help='pairs/s of the N = 1 run on the same box: the line then carries `scaling_efficiency_vs_n1` = '
                    'value / (n_gpus x this).  Or --n1-line FILE with the JSON line of that run')
    ap.add_argument('--n1-line', default=None, help='file holding the JSON line of the N = 1 run on the same box (its last line starting with "{" is read)')
    ap.add_argument('--by-shape', default=None, help='write a per-conv-shape time / TFLOP/s table (debug aid) to this file')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_self_launch(a.gpus))
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (there is no CPU fallback for the HIP path)'
    # (test hooks: RDFC_BENCH_DEVICE pins every rank to one device and RDFC_BENCH_BACKEND=gloo replaces RCCL, so that the N > 1
    #  control flow can be exercised with two ranks on a one-GPU box -- tests/test_dp_gpu.py; RCCL itself needs a GPU per rank)
    if os.environ.get('RDFC_BENCH_DEVICE') is not None:
        local = int(os.environ['RDFC_BENCH_DEVICE'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from rdfc_gan_amd import functional as Fh
    # the compute streams take their hardware queues before RCCL's stream exists (three under data parallelism, four without)
    Fh.reserve_streams(dev, with_net=not (world > 1 or a.rccl_self_test))
    if world > 1 or a.rccl_self_test:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.rccl_self_test:          # one rank, every collective of the step through RCCL all the same (rdfc_gan_amd/optim.py dp_active)
            assert world == 1
            os.environ['RDFC_DP_FORCE'] = '1'
            import socket
            with socket.socket() as sk:                   # (a free rendezvous port: a fixed one collides with a second session on the box)
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            for k, v in (('MASTER_PORT', str(port)), ('RANK', '0'), ('WORLD_SIZE', '1')):
                os.environ.setdefault(k, v)
        backend = os.environ.get('RDFC_BENCH_BACKEND', 'nccl')
        dist.init_process_group(backend, **(dict(device_id=dev) if backend == 'nccl' else {}))
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}: the line would carry the wrong n_gpus'

    from rdfc_gan_amd import functional as Fh
    from rdfc_gan_amd import hip
    from rdfc_gan_amd.rdfc_gan import RDFCGAN
    Fh.set_compute_dtype(dict(bf16=torch.bfloat16, f16=torch.float16, f32=torch.float32)[a.dtype])
    torch.manual_seed(0)
    random.seed(1234 + rank)
    np.random.seed(1234 + rank)
    kw = {}
    if a.full_objective:
        from cases import StandInEncoder
        from rdfc_gan_amd.normal_net import NNET
        a.label_net = True
        kw = dict(normal_net=NNET(a.height, a.width, encoder=StandInEncoder()), out_height=a.height, out_width=a.width)
    model = RDFCGAN(dev, world > 1 or a.rccl_self_test, model_cfg(a.nlspn), is_train=True, exact_schedule=a.exact_schedule, use_label_net=a.label_net, **kw)
    model.train()
    batches = [synth_batch(a.batch, a.height, a.width, 1234 + rank * 100 + i, dev) for i in range(2)]

    def one_step(i, sync=True):
        """the reference's API: optimize_parameters() returns the step's losses as python floats (base.py:121-132,
        train.py:146), i.e. the host waits for them every step -- that is what the headline times"""
        model.set_input(batches[i % len(batches)])
        return model.optimize_parameters(sync_losses=sync and not a.no_sync_losses)

    # Default: eager launches spread over four HIP streams (main, RDFGenerator's depth branch, the B->A half of the G step,
    # weight gradients).  --graph: after one eager warm-up step the step is captured into three HIP graphs (G step | D step |
    # Adam; the data-parallel all-reduces stay eager between them) and replayed.
    graphed = a.graph
    if rank == 0:
        legs = a.dtype == 'f16' and world == 1 and not a.no_as_worded and not graphed and not a.label_net and not a.nlspn
        _log(f'headline: {a.warmup} warm-up + {a.steps} timed steps'
             + ('; then extra legs (bf16 as worded, #3 NLSPN, #5 480x640: ~1 minute)' if legs else '')
             + ('' if a.no_cpu_baseline or world > 1 else f'; then the CPU baseline (--cpu-steps {a.cpu_steps}: ~{2 * max(a.cpu_steps, 1) + 0.5:.0f} minutes of host time)'))
    for i in range(a.warmup):
        if graphed and i == 1:
            model.enable_graph(True)
        # the last warm-up step carries kernel events like every --event-every-th timed step: the runtime grows its event /
        # signal pools on first use, which would otherwise land in the first timed step (seen as rare 40 ms outliers)
        warm_ev = not graphed and not a.no_kernel_events and i == a.warmup - 1
        hip.PROFILE = [] if warm_ev else None
        hip.PROFILE_HBM = [] if warm_ev else None
        one_step(i)
        hip.PROFILE = hip.PROFILE_HBM = None
    if graphed and a.warmup < 2:
        model.enable_graph(True)
        one_step(a.warmup)                 # capture + first replay (untimed)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Per-kernel HIP events (roofline), inside the timed region: an event pair around every conv launch of every
    # --event-every-th timed step (each pair costs ~3 us of host time; the step is launch-bound, so they are sampled).
    # A replayed graph cannot carry torch events: with --graph the LAST timed step runs eagerly with them instead.
    events = not a.no_kernel_events
    prof_all, prof_hbm, ev_steps = [], [], 0
    t0 = time.perf_counter()
    for i in range(a.steps):
        with_ev = events and (i == a.steps - 1 if graphed else i % max(1, a.event_every) == 0)
        if with_ev:
            if graphed:
                model.enable_graph(False)
            hip.PROFILE, hip.PROFILE_HBM = prof_all, prof_hbm
            ev_steps += 1
        else:
            hip.PROFILE = hip.PROFILE_HBM = None
        losses = one_step(i)
    enqueue = time.perf_counter() - t0          # host time to enqueue the K steps (== elapsed when launch-bound or synced)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof, hip.PROFILE, hip.PROFILE_HBM = prof_all, None, None
    if isinstance(next(iter(losses.values())), float):
        final = losses
    else:
        final = model.reduce_loss(losses)
    # extra (untimed by the contract): the same K steps without the per-step loss read-back -- the host runs ahead of the GPU
    unsynced = None
    if not a.no_sync_losses:
        t1 = time.perf_counter()
        for i in range(a.steps):
            one_step(i, sync=False)
        enq1 = time.perf_counter() - t1
        torch.cuda.synchronize()
        unsynced = (time.perf_counter() - t1, enq1)
    # host cost of enqueueing ONE step into an idle queue (no back-pressure from earlier steps): median of three
    host_one = []
    for i in range(3):
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        one_step(i, sync=False)
        host_one.append(time.perf_counter() - t2)
    torch.cuda.synchronize()
    host_one = sorted(host_one)[1]
    ev_steps = max(1, ev_steps)                                   # timed steps that carried per-kernel events
    # Untimed extra step on ONE stream with the same events: in the timed steps a kernel shares the chip with the other
    # streams' kernels, so its duration there (what rocprofv3 of this command shows too) understates what it does alone.
    iso, iso_hbm = [], []
    if events:                          # on every rank: the step contains collectives
        if graphed:
            model.enable_graph(False)
        Fh.set_multi_stream(False)
        hip.PROFILE, hip.PROFILE_HBM = iso, iso_hbm
        one_step(a.steps)
        torch.cuda.synchronize()
        hip.PROFILE = hip.PROFILE_HBM = None
        Fh.set_multi_stream(True)
    rmse_dev = rmse_deviation(model, a.height, a.width, dev) if rank == 0 else None     # (untimed; no collective in the eval path)
    comm = None
    if world > 1 or a.rccl_self_test:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        # self-check of the exchange (untimed): which backend / world the collectives really run on and what one all-reduce
        # of every gradient bucket costs alone, so that a scaling curve can be read against it
        comm = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), buckets=[])
        if dist.get_backend() == 'nccl':
            comm['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        comm['env'] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(('NCCL_', 'RCCL_', 'HSA_ENABLE_IPC', 'GPU_MAX_HW_QUEUES'))}
        comm.update(comm_in_step(model, one_step, a.steps, dev))
        for oname, o in model.optimizers.items():
            for bi in range(len(o._buckets)):
                views = [o.flat_grad[lo:hi] for lo, hi in o.bucket_ranges(bi)]      # (a bucket is one or a few ranges of the flat gradient)
                nel = sum(v.numel() for v in views)
                for v in views:
                    dist.all_reduce(v)                                 # warm-up (RCCL sets up its rings per size class)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for v in views:
                    dist.all_reduce(v)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                comm['buckets'].append(dict(optimizer=oname, bucket=bi, ranges=len(views), mbytes=round(4e-6 * nel, 1), allreduce_ms=round(ms, 3),
                                            bus_gb_s=round(2 * (world - 1) / world * 4e-9 * nel / (ms * 1e-3), 1)))
    assert all(np.isfinite(v) for v in final.values()), f'non-finite losses: {final}'

    if rank == 0:
        scale = (a.height * a.width) / (228.0 * 304.0)
        gflop_pair = step_gflop_per_pair(a.exact_schedule, scale, a.label_net)
        pairs_s = world * a.batch * a.steps / elapsed
        out = dict(metric='RGB-D pairs/sec per train step (G+D), NYUv2 304x228', value=round(pairs_s, 3), unit='pairs/s',
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * elapsed / a.steps, 3),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype=a.dtype, data='synthetic',
                   config=dict(workload=f'RDFCGAN.optimize_parameters: G_A2B(RDFGenerator, NLSPN {"on" if a.nlspn else "off"}) + '
                                        f'G_B2A(ResnetGenerator IN2d) + D_A + D_B (PatchGAN){" + G_label (PSPNet-ResNet50) x2 + label / normal terms" if a.label_net else ""}{" + G_normal (decoder; stand-in encoder) x3" if a.full_objective else ""}, fwd+bwd+Adam, {a.height}x{a.width}, '
                                        f'batch {a.batch}/GPU, random-init weights',
                               global_batch=world * a.batch, parallelism=f'dp{world}',
                               dead_forwards='executed' if a.exact_schedule else 'elided (result-identical, see DESIGN.md)',
                               step_gflop_per_pair=round(gflop_pair, 1)),
                   loss_readback='every step (reference API)' if not a.no_sync_losses else 'none in the timed region',
                   host_enqueue_ms_per_step=round(1e3 * host_one, 3),
                   unsynced_ms_per_step=round(1e3 * unsynced[0] / a.steps, 3) if unsynced else None, launch='eager, 4 streams (main | depth branch, then the discriminators\' step | weight gradients | D_B\'s G-loss pass), batched calls' if not graphed else 'hip graphs (3 per step), one stream',
                   step_tflops=round(pairs_s * gflop_pair / 1e3, 2),
                   step_frac_of_mfma_peak=round(pairs_s * gflop_pair / 1e3 / (PEAK_TFLOPS[a.dtype] * world), 4))
        # multi-GPU reading aids (the driver computes scaling efficiency itself from the per-N `value`s; these only make one line self-explanatory)
        out['value_per_gpu'] = round(pairs_s / world, 3)
        n1 = a.n1_value
        if n1 is None and a.n1_line:
            try:
                n1 = float(json.loads([l for l in open(a.n1_line).read().splitlines() if l.startswith('{')][-1])['value'])
            except (OSError, IndexError, KeyError, ValueError) as e:
                out['scaling_efficiency_note'] = f'--n1-line {a.n1_line}: unreadable ({e})'
        if n1:
            out['scaling_efficiency_vs_n1'] = round(pairs_s / (world * n1), 4)
            out['n1_value'] = n1
        out['rmse_dev_m'] = round(rmse_dev, 6)        # eval-path dense-depth RMSE of this dtype against the f32 parity mode, metres (north star: <= 1e-3)
        if model.scaler is not None:
            out['loss_scale'] = dict(scale=model.scaler.scale(), dynamic=model.scaler.dynamic, skipped_steps=model.scaler.skipped())
        if comm is not None:
            out['comm'] = comm
        if prof:
            fam, sym = {}, {}
            for name, flops, e0, e1, _shape, kernel in prof:
                ms = e0.elapsed_time(e1)
                for tab, key in ((fam, name), (sym, kernel)):
                    d = tab.setdefault(key, [0.0, 0.0, 0])
                    d[0] += flops
                    d[1] += ms * 1e-3
                    d[2] += 1
            # the dominant kernel = the single kernel symbol (template instantiation, as rocprofv3 names it) with the
            # largest summed duration over the timed region; algorithmic flops = 2*N*Ho*Wo*Co*Ci*KH*KW per launch
            top = max(sym, key=lambda k: sym[k][1])
            fl, sec, n = sym[top]
            peak = PEAK_TFLOPS[a.dtype]
            out['roofline'] = dict(bound='mfma', kernel=top, achieved=round(fl / sec / 1e12, 2), peak=peak, unit='TFLOP/s',
                                   frac=round(fl / sec / 1e12 / peak, 4), traffic=None, launches=n,
                                   avg_launch_us=round(1e6 * sec / n, 2),
                                   share_of_step=round(sec / (elapsed * ev_steps / a.steps), 4),
                                   event_steps=ev_steps)
            # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from the
            # committed rocprofv3 passes over this same command (tools/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in separate
            # passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), apply to the default workload only and
            # are refused when they were taken from other kernel sources than the ones this library was built from
            default_wl = (a.batch, a.height, a.width, a.nlspn, a.label_net, a.exact_schedule) == (8, 228, 304, False, False, False)   # (--full-objective sets label_net)
            suffix = {'f16': '', 'bf16': '_bf16'}.get(a.dtype) if default_wl else ('_nlspn' if (a.batch, a.height, a.width, a.dtype, a.nlspn, a.label_net, a.exact_schedule) == (8, 228, 304, 'f16', True, False, False) else None)
            if suffix is not None:
                meta = pmc_for_kernel(out['roofline'], top, suffix, hip.build_hash())
                if meta is not None and meta.get('fabric_gb_per_step') is not None:
                    # every kernel's (FETCH_SIZE x 2 + WRITE_SIZE) x launches of the profiled steps / those steps: what a step moves between the
                    # L2s and the fabric (Infinity Cache + HBM); by category in the summary's _meta
                    out['fabric_gb_per_step'] = meta['fabric_gb_per_step']
                    out['fabric_gb_per_step_by_category'] = meta.get('fabric_gb_by_category')
            if iso:
                ifl = sum(r[1] for r in iso if r[5] == top)
                isec = sum(r[2].elapsed_time(r[3]) * 1e-3 for r in iso if r[5] == top)
                inum = sum(1 for r in iso if r[5] == top)
                if inum:
                    out['roofline']['isolated'] = dict(achieved=round(ifl / isec / 1e12, 2), frac=round(ifl / isec / 1e12 / peak, 4),
                                                       avg_launch_us=round(1e6 * isec / inum, 2), launches=inum,
                                                       note='same kernel on one extra untimed single-stream step (nothing else on the chip): what the kernel '
                                                            'does alone.  In the timed steps it shares the chip with the HBM-bound normalisation kernels and the '
                                                            'weight gradients of two other streams, which is what `achieved` above and the rocprofv3 summary of '
                                                            'this command (profiles/) measure')
            out['kernels'] = {k: dict(tflops=round(v[0] / v[1] / 1e12, 2), avg_launch_us=round(1e6 * v[1] / v[2], 2),
                                      launches=v[2], share_of_step=round(v[1] / (elapsed * ev_steps / a.steps), 4))
                              for k, v in sorted(sym.items(), key=lambda kv: -kv[1][1])[:8]}
            out['kernel_families'] = {k: dict(tflops=round(v[0] / v[1] / 1e12, 2), seconds=round(v[1], 4), launches=v[2],
                                              share_of_step=round(v[1] / (elapsed * ev_steps / a.steps), 4)) for k, v in sorted(fam.items())}
        if prof_hbm:
            # HBM-bound kernels: algorithmic bytes per launch / the launch's duration (HIP events on its stream, inside the
            # timed steps that carried events); `isolated`: the same on the untimed single-stream step
            def hbm_table(rows):
                tab = {}
                for kernel, nbytes, e0, e1, launches in rows:
                    d = tab.setdefault(kernel, [0.0, 0.0, 0])
                    d[0] += nbytes
                    d[1] += e0.elapsed_time(e1) * 1e-3
                    d[2] += launches
                return tab
            ht, hi = hbm_table(prof_hbm), hbm_table(iso_hbm)
            keep = sorted(ht, key=lambda k: -ht[k][1])[:3] + [k for k in ht if k == 'prop_fwd']
            out['roofline_hbm'] = []
            for k in dict.fromkeys(keep):
                nb, sec, n = ht[k]
                e = dict(kernel=k, bound='hbm', launches=n, algorithmic_bytes_per_launch=round(nb / n), avg_launch_us=round(1e6 * sec / n, 2),
                         achieved=round(nb / sec / 1e9, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(nb / sec / 1e9 / PEAK_HBM_GBS, 4),
                         share_of_step=round(sec / (elapsed * ev_steps / a.steps), 4))
                if k in hi and hi[k][1] > 0:
                    e['isolated'] = dict(achieved=round(hi[k][0] / hi[k][1] / 1e9, 1), frac=round(hi[k][0] / hi[k][1] / 1e9 / PEAK_HBM_GBS, 4),
                                         avg_launch_us=round(1e6 * hi[k][1] / hi[k][2], 2))
                out['roofline_hbm'].append(e)
        if prof and a.by_shape:
            tab = {}
            for name, flops, e0, e1, shp, kernel in prof:
                d = tab.setdefault((kernel.replace(' ', ''),) + shp, [0.0, 0.0, 0])
                d[0] += flops
                d[1] += e0.elapsed_time(e1) * 1e-3
                d[2] += 1
            with open(a.by_shape, 'w') as f:
                f.write('kernel N Hi Wi Ci Ho Wo Co K s | launches ms_total us_avg TFLOP/s\n')
                for k, v in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                    f.write(' '.join(map(str, k)) + f' | {v[2]} {1e3 * v[1]:.2f} {1e6 * v[1] / v[2]:.1f} {v[0] / v[1] / 1e12:.1f}\n')
        if a.dtype == 'f16' and world == 1 and not a.no_as_worded and not graphed and not a.label_net and not a.nlspn:
            # The other single-GPU configurations of BASELINE.json, measured in the same process on the same box after the headline
            # (extras, untimed by the contract): configs[1] AS WORDED (bf16 conv operands) with its own roofline block, configs[2]
            # (= the headline + NLSPN propagation) and configs[4]'s per-GPU shape (480 x 640, batch 4, f16).  Same protocol as the
            # headline (the same warm-up count, losses read back every step), min(steps, 10) timed steps each.
            del model
            torch.cuda.empty_cache()

            def leg(dtype, nlspn, batch, H, W, tag, pmc_suffix):
                _log(f'extra leg: {tag[:60]} ...')
                Fh.set_compute_dtype(dict(bf16=torch.bfloat16, f16=torch.float16)[dtype])
                Fh.invalidate_packed_weights()
                torch.manual_seed(0)
                m = RDFCGAN(dev, False, model_cfg(nlspn), is_train=True, exact_schedule=a.exact_schedule)
                m.train()
                bl = batches if (batch, H, W) == (a.batch, a.height, a.width) else [synth_batch(batch, H, W, 4321 + i, dev) for i in range(2)]
                nst = min(a.steps, 10)
                for i in range(a.warmup):
                    m.set_input(bl[i % 2])
                    m.optimize_parameters()
                torch.cuda.synchronize()
                lprof, liso = [], []
                tb = time.perf_counter()
                for i in range(nst):
                    hip.PROFILE = lprof if i == 0 else None          # per-kernel HIP events on the first timed step (as the headline)
                    m.set_input(bl[i % 2])
                    ll = m.optimize_parameters()
                torch.cuda.synchronize()
                el = time.perf_counter() - tb
                hip.PROFILE = None
                assert all(np.isfinite(v) for v in ll.values()), f'non-finite losses in leg {tag}: {ll}'
                Fh.set_multi_stream(False)                           # one extra untimed step on ONE stream: every kernel alone on the chip
                hip.PROFILE = liso
                m.set_input(bl[0])
                m.optimize_parameters()
                torch.cuda.synchronize()
                hip.PROFILE = None
                Fh.set_multi_stream(True)
                sc = (H * W) / (228.0 * 304.0)
                gf = step_gflop_per_pair(a.exact_schedule, sc, False)
                r = dict(value=round(batch * nst / el, 3), unit='pairs/s', ms_per_step=round(1e3 * el / nst, 3), steps=nst, warmup=a.warmup, dtype=dtype,
                         config=dict(workload=tag, batch=batch, height=H, width=W, nlspn=bool(nlspn), step_gflop_per_pair=round(gf, 1)),
                         step_frac_of_mfma_peak=round(batch * nst / el * gf / 1e3 / PEAK_TFLOPS[dtype], 4))
                sym = {}
                for _n, fl_, e0, e1, _s, kern in lprof:
                    d_ = sym.setdefault(kern, [0.0, 0.0, 0])
                    d_[0] += fl_
                    d_[1] += e0.elapsed_time(e1) * 1e-3
                    d_[2] += 1
                if sym:
                    top_ = max(sym, key=lambda k: sym[k][1])
                    fl_, sec_, n_ = sym[top_]
                    r['roofline'] = dict(bound='mfma', kernel=top_, achieved=round(fl_ / sec_ / 1e12, 2), peak=PEAK_TFLOPS[dtype], unit='TFLOP/s',
                                         frac=round(fl_ / sec_ / 1e12 / PEAK_TFLOPS[dtype], 4), traffic=None, launches=n_,
                                         avg_launch_us=round(1e6 * sec_ / n_, 2), share_of_step=round(sec_ / (el / nst), 4), event_steps=1)
                    ifl = sum(x[1] for x in liso if x[5] == top_)
                    isec = sum(x[2].elapsed_time(x[3]) * 1e-3 for x in liso if x[5] == top_)
                    inum = sum(1 for x in liso if x[5] == top_)
                    if inum:
                        r['roofline']['isolated'] = dict(achieved=round(ifl / isec / 1e12, 2), frac=round(ifl / isec / 1e12 / PEAK_TFLOPS[dtype], 4),
                                                         avg_launch_us=round(1e6 * isec / inum, 2), launches=inum)
                    lmeta = pmc_for_kernel(r['roofline'], top_, pmc_suffix, hip.build_hash())
                    if lmeta is not None and lmeta.get('fabric_gb_per_step') is not None:
                        r['fabric_gb_per_step'] = lmeta['fabric_gb_per_step']
                del m
                torch.cuda.empty_cache()
                return r
            out['as_worded_bf16'] = leg('bf16', False, a.batch, a.height, a.width, 'BASELINE configs[1] as worded: the headline workload with bf16 conv operands', '_bf16')
            out['as_worded_bf16']['note'] = 'same process, same box, same schedule and protocol as the headline, after the f16 measurement'
            if (a.batch, a.height, a.width) == (8, 228, 304):
                out['configs'] = {
                    '#3_nlspn': leg('f16', True, 8, 228, 304, 'BASELINE configs[2]: configs[1] + NLSPN propagation (18 steps, TGASS, confidence propagation; f32 gather kernels)', '_nlspn'),
                    '#5_480x640_b4_f16': leg('f16', False, 4, 480, 640, "BASELINE configs[4]'s per-GPU shape: SUN RGB-D 640x480 inputs, batch 4, f16 activations, f32 loss accumulation", '_480x640'),
                }
            Fh.set_compute_dtype(torch.float16)
        if not a.no_cpu_baseline and world == 1:    # rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(a.batch, a.height, a.width, True, steps=a.cpu_steps)
    if world > 1 or a.rccl_self_test:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line on stdout: RCCL prints a version banner through C stdio, which -- block-buffered on a
        # pipe -- would otherwise come out at exit, behind it
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()

"""CPU, world_size 2 over gloo: the data-parallel host logic of the train step -- one flat-bucket gradient
all-reduce per optimizer, DDP-style initial broadcast, packed loss reduction.  (The arithmetic kernels need a
GPU; here only the communication pattern and the flat-buffer bookkeeping are exercised.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from rdfc_gan_amd.optim import FlatAdam, broadcast_module_state
        from rdfc_gan_amd.rdfc_gan import RDFCGAN
        torch.manual_seed(100 + rank)                       # different init per rank ...
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
        broadcast_module_state([net])                       # ... made identical, buffers included
        ref = [t.clone() for t in list(net.parameters()) + list(net.buffers())]
        gathered = [None] * world
        dist.all_gather_object(gathered, [t.tolist() for t in ref])
        assert gathered[0] == gathered[1], 'broadcast_module_state left ranks different'
        opt = FlatAdam(net.parameters(), lr=1e-3)
        # parameters and grads are views of the flat buffers
        assert all(p.data_ptr() >= opt.flat.data_ptr() for p in net.parameters())
        for p in net.parameters():
            p.grad.fill_(float(rank + 1))                   # rank-dependent gradient written through the views
        opt.all_reduce_grads_async()                        # gloo: synchronous sum into the flat bucket
        opt.wait_grads()
        assert torch.all(opt.flat_grad == 3.0), 'flat gradient bucket is not the sum over ranks'
        # buckets: element ranges reduced separately, in any order, each exactly once per step
        opt.zero_grad()
        lo, hi = opt.param_range(list(net[2].parameters()))
        assert hi == opt.flat.numel() and 0 < lo < hi
        opt.set_buckets([(lo, hi), (0, lo)])
        opt.flat_grad.fill_(float(rank + 1))
        opt.all_reduce_bucket_async(0)
        assert torch.all(opt.flat_grad[lo:] == 3.0) and torch.all(opt.flat_grad[:lo] == float(rank + 1))
        opt.all_reduce_grads_async()                        # the rest; bucket 0 is not reduced a second time
        opt.wait_grads()
        assert torch.all(opt.flat_grad == 3.0), 'a bucket was skipped or reduced twice'
        # a bucket made of several ranges (parameters that become final together need not be neighbours: G_A2B's decoder side sits
        # between its encoders in parameter order): both ranges go out with the bucket, the issue order is recorded
        opt.zero_grad()
        l0, h0 = opt.param_range(list(net[0].parameters()))
        l1, h1 = opt.param_range(list(net[1].parameters()))
        assert (l0, h1, h0) == (0, lo, l1)
        opt.set_buckets([[(l0, h0), (lo, hi)], [(l1, h1)]])
        assert opt.bucket_ranges(0) == [(l0, h0), (lo, hi)] and opt.bucket_ranges(1) == [(l1, h1)]
        opt.flat_grad.fill_(float(rank + 1))
        opt.all_reduce_bucket_async(0)
        assert torch.all(opt.flat_grad[l0:h0] == 3.0) and torch.all(opt.flat_grad[lo:hi] == 3.0) and torch.all(opt.flat_grad[l1:h1] == float(rank + 1))
        opt.all_reduce_grads_async()
        opt.wait_grads()
        assert torch.all(opt.flat_grad == 3.0) and opt.started_order == [0, 1]
        opt.set_buckets([(0, opt.flat.numel())])
        # autograd accumulates into the persistent views (no re-allocation of .grad)
        opt.zero_grad()
        x = torch.randn(2, 3, 6, 6)
        net(x).sum().backward()
        assert float(opt.flat_grad.abs().sum()) > 0 and all(
            opt.flat_grad.data_ptr() <= p.grad.data_ptr() < opt.flat_grad.data_ptr() + 4 * opt.flat_grad.numel()
            for p in net.parameters())
        # the optimizer itself is a HIP kernel: it must refuse to run on CPU rather than fall back
        try:
            opt.step()
            raise AssertionError('FlatAdam.step silently ran on CPU')
        except RuntimeError as e:
            assert 'no CPU fallback' in str(e)
        # packed loss reduction == mean over ranks, one collective
        stats = dict(a=torch.tensor(float(rank)), b=torch.tensor(10.0 * (rank + 1)))
        red = RDFCGAN.reduce_loss(None, stats)
        assert abs(red['a'] - 0.5) < 1e-6 and abs(red['b'] - 15.0) < 1e-6, red
        # the step's loss read-back: groups posted as they become known, ONE averaging collective when they are collected,
        # keys back in the order asked for
        from rdfc_gan_amd.rdfc_gan import _LossReadback
        rb = _LossReadback(torch.device('cpu'))
        rb.begin()
        rb.post(dict(loss_G=torch.tensor(1.0 + rank), loss_cycle_A=torch.tensor(2.0)))
        rb.post(dict(loss_D_A=torch.tensor(4.0 * rank)))
        got = rb.collect(['loss_D_A', 'loss_G', 'loss_cycle_A'])
        assert list(got) == ['loss_D_A', 'loss_G', 'loss_cycle_A'] and got == dict(loss_D_A=2.0, loss_G=1.5, loss_cycle_A=2.0), got
        assert all(isinstance(v, float) for v in got.values())
        # optimizer state round-trips in torch.optim.Adam's checkpoint format
        sd = opt.state_dict()
        assert set(sd) == {'state', 'param_groups'} and set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
        opt.load_state_dict(sd)
        q.put((rank, 'ok'))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_host_logic_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', f'rank {rank}: {msg}'

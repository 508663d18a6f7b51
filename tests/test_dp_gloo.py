"""Data-parallel semantics with world_size 2 over gloo on the CPU: per-rank gradients of group-
preserving shards, summed by all-reduce and scaled by 1/world, equal the global-batch gradient.
The compute here is the CPU oracle (the HIP kernels need a GPU); sharding and reduction are the
package's."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from helpers import oracle_params
    from eeggan_amd import config
    from eeggan_amd.parallel import shard_batch
    from oracle import ref_cpu as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    depth, alpha, B = 1, 0.5, 16
    g, d = config.build_models(3)
    gp, gs = oracle_params(g)
    dp, ds = oracle_params(d)
    for p in dp.values():
        p.requires_grad_(True)
    gen = torch.Generator().manual_seed(0)
    real = torch.randn(B, 5, config.seq_len(depth), generator=gen, dtype=torch.float64)
    z = torch.randn(B, 256, generator=gen, dtype=torch.float64)
    mix = torch.rand(B, 1, 1, generator=gen, dtype=torch.float64)
    cfg = O.NetConfig(g_dropout=0.0, d_dropout=0.0)  # noise off: it is drawn per rank
    off = O.DropSource("off")

    def grads(r, zz, m):
        loss, _ = O.d_loss_wgan_gp(dp, ds, gp, gs, r, zz, depth, alpha, cfg, m, off, off)
        return loss.detach(), torch.autograd.grad(loss, list(dp.values()), allow_unused=True)

    loss_l, local = grads(shard_batch(real, rank, world), shard_batch(z, rank, world), shard_batch(mix, rank, world))
    flat = torch.cat([g_.reshape(-1) for g_ in local if g_ is not None])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    dist.all_reduce(loss_l, op=dist.ReduceOp.SUM)
    if rank == 0:
        loss_g, full = grads(real, z, mix)
        ref = torch.cat([g_.reshape(-1) for g_ in full if g_ is not None])
        out.put((float((flat - ref).norm() / ref.norm()), float(loss_l / world), float(loss_g)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_global_batch():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    err, loss_avg, loss_full = out.get()
    assert err < 1e-10, err
    assert abs(loss_avg - loss_full) < 1e-10

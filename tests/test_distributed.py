"""N>1 path on CPU: two gloo ranks exercise the gradient all-reduce, parameter broadcast and batch sharding helpers
that bench.py / training.train_step use over RCCL on the GPU box."""
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from mpnn_amd import distributed

    r, _, w = distributed.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # ranks start from different weights ...
    lin = torch.nn.Linear(5, 3)
    extra = torch.nn.Parameter(torch.randn(7))  # ... and one parameter never receives a gradient on rank 1
    params = list(lin.parameters()) + [extra]
    distributed.broadcast_parameters(params, src=0)
    w0 = torch.cat([p.detach().reshape(-1) for p in params]).clone()
    x = torch.full((4, 5), float(rank + 1))
    loss = lin(x).sum() + (extra.sum() if rank == 0 else 0.0)
    loss.backward()
    distributed.allreduce_gradients(params, world)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    # every gradient is now a view of ONE flat buffer (no copy back), in parameter order
    base = params[0].grad.untyped_storage().data_ptr()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in params)
    assert params[1].grad.data_ptr() == params[0].grad.data_ptr() + 4 * params[0].numel()
    # a second step: fresh gradients from autograd, same buffer, same averages
    for p in params:
        p.grad = None
    loss = lin(x).sum() + (extra.sum() if rank == 0 else 0.0)
    loss.backward()
    distributed.allreduce_gradients(params, world)
    assert params[0].grad.untyped_storage().data_ptr() == base
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in params]), flat)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    gw = [torch.empty_like(w0) for _ in range(world)]
    dist.all_gather(gw, w0)
    if rank == 0:
        # plain lists: tensors would travel as shared-memory handles that die with this process
        out.put((flat.tolist(), [g.tolist() for g in gathered], [g.tolist() for g in gw]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_and_broadcast_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, gathered, weights = q.get(timeout=120)
    flat = torch.tensor(flat)
    gathered = [torch.tensor(g) for g in gathered]
    weights = [torch.tensor(w) for w in weights]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(weights[0], weights[1]), "broadcast: both ranks hold rank 0's weights"
    assert torch.equal(gathered[0], gathered[1]), "all ranks end with the same averaged gradient"
    # d/dW of sum(W x + b) with x = (rank+1) * ones(4,5): 4 (rank+1) per weight, 4 per bias; mean over ranks
    assert torch.allclose(flat[:15], torch.full((15,), 4 * 1.5))
    assert torch.allclose(flat[15:18], torch.full((3,), 4.0))
    assert torch.allclose(flat[18:], torch.full((7,), 0.5)), "a missing gradient counts as zero in the mean"


def test_shard_range_partitions_exactly():
    from mpnn_amd.distributed import shard_range

    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1

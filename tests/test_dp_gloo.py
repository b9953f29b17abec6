"""CPU tier: the data-parallel gradient exchange (train.DataParallel) with world_size 2 on gloo.
Checks the sliced all-reduce over the flat gradient buffer, the folded 1/world scale and the
parameter/buffer broadcast."""
import os
import socket
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_gan_amd.train import DataParallel
    dp = DataParallel(slice_bytes=4096)            # 1024-element slices -> several all-reduces
    n = 5000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    opt = types.SimpleNamespace(flat_g=g)
    scale = dp.reduce(opt)
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(opt.flat_g, want) and abs(scale - 1.0 / world) < 1e-12
    model = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        for p in model.parameters():
            p.fill_(float(rank + 1))
        model[1].running_mean.fill_(float(rank + 7))
    dp.broadcast_model(model)
    ok = ok and all(bool((p == 1.0).all()) for p in model.parameters()) and bool((model[1].running_mean == 7.0).all())
    out[rank] = ok
    dist.destroy_process_group()


def test_dp_reduce_and_broadcast_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_single_process_is_identity():
    from vae_gan_amd.train import DataParallel
    dp = DataParallel()
    opt = types.SimpleNamespace(flat_g=torch.ones(8))
    assert dp.reduce(opt) == 1.0 and dp.world == 1

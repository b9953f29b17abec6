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


def _worker(rank, world, port, out, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_gan_amd.train import DataParallel
    dp = DataParallel(slice_bytes=4096, exchange=exchange)     # 1024-element slices -> several exchanges
    n = 5001                                       # odd: the last bucket has a remainder the shards do not cover
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


import pytest


@pytest.mark.parametrize("exchange,world", [("allreduce", 2), ("rsag", 2), ("rsag", 3)])
def test_dp_reduce_and_broadcast(exchange, world):
    """Both exchanges (VG_DP_EXCHANGE): one all-reduce per bucket, or reduce-scatter + all-gather on the bucket."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, exchange), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_dp_exchange_switch_from_environment(monkeypatch):
    from vae_gan_amd.train import DataParallel
    monkeypatch.setenv("VG_DP_EXCHANGE", "rsag")
    assert DataParallel().exchange == "rsag"
    monkeypatch.setenv("VG_DP_EXCHANGE", "ring")
    with pytest.raises(ValueError):
        DataParallel()


def test_single_process_is_identity():
    from vae_gan_amd.train import DataParallel
    dp = DataParallel()
    opt = types.SimpleNamespace(flat_g=torch.ones(8))
    assert dp.reduce(opt) == 1.0 and dp.world == 1


def _worker_buckets(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_gan_amd.train import DataParallel
    dp = DataParallel(bucket_bytes=64)             # 16-element pieces
    # three parameter groups laid out like FlatAdam does (gradients are views into one flat buffer)
    sizes = [40, 24, 56]
    params = [torch.nn.Parameter(torch.full((n,), 0.5 + i)) for i, n in enumerate(sizes)]
    offsets = [0, 40, 64]
    flat_g = torch.zeros(sum(sizes))
    for p, o in zip(params, offsets):
        p.grad = flat_g[o:o + p.numel()]
    opt = types.SimpleNamespace(params=params, offsets=offsets, numel=flat_g.numel(), flat_g=flat_g)
    x = torch.full((8,), float(rank + 1), requires_grad=True)
    h = x * params[0][:8].sum()                       # "encoder": group 0
    fired = []
    h = dp.mark(h, opt, params[1:])                   # "decoder": groups 1 and 2, exchanged from the boundary
    orig_send = dp._send
    dp._send = lambda o, lo, hi: (fired.append((lo, hi)), orig_send(o, lo, hi))[1]
    y = (h * params[1][:8]).sum() + (h.sum() * params[2]).sum()
    y.backward()
    after_backward = list(fired)
    scale = dp.reduce(opt)
    # expected: the same graph evaluated for every rank's x, gradients added up
    want = torch.zeros_like(flat_g)
    for r in range(world):
        ps = [torch.nn.Parameter(p.detach().clone()) for p in params]
        xr = torch.full((8,), float(r + 1))
        hr = xr * ps[0][:8].sum()
        ((hr * ps[1][:8]).sum() + (hr.sum() * ps[2]).sum()).backward()
        for p, o in zip(ps, offsets):
            want[o:o + p.numel()] += p.grad
    out[rank] = (after_backward == [(40, 120)] and fired == [(40, 120), (0, 40)] and dp.n_async == 1
                 and torch.allclose(flat_g, want) and abs(scale - 1.0 / world) < 1e-12
                 and DataParallel.span(opt, params[:2]) == (0, 64))
    dist.destroy_process_group()


def test_dp_bucket_boundaries_world2():
    """A boundary fires from inside the backward pass once its group's gradients exist; reduce() sends the rest."""
    world = 2
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_buckets, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def _worker_stats(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_gan_amd import ops
    from vae_gan_amd.train import DataParallel
    dp = DataParallel(bn_sync="global")
    ok = ops.STAT_SYNC[0] is dp
    t = torch.full((6,), float(rank + 1), dtype=torch.float64)
    dp.sum_(t)
    ok = ok and bool((t == sum(r + 1 for r in range(world))).all())
    dp.close()
    out[rank] = ok and ops.STAT_SYNC[0] is None
    dist.destroy_process_group()


def test_dp_global_statistics_hook_world2():
    world = 2
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_stats, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_device_loader_reshuffles_every_epoch_with_world_gt_1(tmp_path):
    """under torchrun every rank walked its shard in the same order in every epoch (nothing called
    DistributedSampler.set_epoch).  DeviceLoader.set_epoch, called by loop.train at the top of each epoch, makes the
    order a function of the epoch, like the single-process loader's shuffle=True and the reference's (data.py:133-139)."""
    from PIL import Image
    import numpy as np
    from vae_gan_amd import data as D
    d = tmp_path / "kodak"
    d.mkdir()
    for i in range(12):
        Image.fromarray(np.full((8, 8, 3), i, dtype=np.uint8)).save(d / f"kodim{i:02d}.png")

    class Cfg:
        data_dir, dataset, image_size, seed, batch_size, eval_batch_size, num_workers = str(tmp_path), "kodak", 8, 42, 2, 1, 0

    loader = D.get_dataloader(Cfg(), "train", rank=1, world=2)
    loader.transform = lambda images: [int(im[0, 0, 0]) for im in images]       # no GPU here: just which files came
    orders = []
    for epoch in range(2):
        loader.set_epoch(epoch)
        orders.append([i for batch in loader for i in batch])
    assert sorted(orders[0]) != list(range(12)) and len(orders[0]) == 6         # a shard of half the files
    assert orders[0] != orders[1], "the same order in both epochs"
    loader.set_epoch(0)
    assert [i for batch in loader for i in batch] == orders[0]                 # reproducible per (seed, epoch)

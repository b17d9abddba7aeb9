"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange used by bench.py / the
trainer for N > 1 GPUs: one all-reduce of the packed gradient buffer, replicas stay identical."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khairil_tum_facade_semantic_segmentation_amd.train import FlatGradients
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    fg = FlatGradients(net)
    opt = torch.optim.Adam(fg.params, lr=1e-2)
    xs = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 6) / 10.0
    x = xs[rank::world]                                    # each rank its own shard of the global batch
    fg.zero()
    loss = net(x).pow(2).sum()
    loss.backward()
    fg.all_reduce_mean()
    flat = fg.buffer.clone()
    opt.step()
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    # single-process reference over the full batch: mean of per-rank sums == sum / world
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    (ref(xs).pow(2).sum() / world).backward()
    ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    ok = torch.allclose(flat, ref_flat, rtol=1e-5, atol=1e-6) and all(torch.equal(gathered[0], g) for g in gathered)
    ok = ok and all(p.grad.data_ptr() >= fg.buffer.data_ptr() for p in fg.params)      # grads are views of the buffer
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_world2():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_single_process_is_a_noop():
    from khairil_tum_facade_semantic_segmentation_amd.train import FlatGradients
    net = torch.nn.Linear(3, 2)
    fg = FlatGradients(net)
    fg.zero()
    assert all(p.grad is None for p in fg.params)
    net(torch.ones(1, 3)).sum().backward()
    fg.all_reduce_mean()                                    # no process group: nothing to do
    assert fg.buffer is None and fg.numel == 8


def test_rotate_z_matches_reference_formula():
    """provider.rotate_point_cloud_z (provider.py:66-84): rotated = pc @ [[c,s,0],[-s,c,0],[0,0,1]]."""
    import numpy as np
    from khairil_tum_facade_semantic_segmentation_amd.train import rotate_z_
    rs = np.random.RandomState(1)
    pc = rs.normal(size=(3, 50, 9)).astype(np.float32)
    ang = rs.uniform(0, 2 * np.pi, size=3)
    want = pc.copy()
    for k in range(3):
        c, s = np.cos(ang[k]), np.sin(ang[k])
        R = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
        want[k, :, :3] = pc[k, :, :3].reshape(-1, 3) @ R
    got = rotate_z_(torch.from_numpy(pc).permute(0, 2, 1).contiguous(), torch.from_numpy(ang).float())
    assert np.abs(got.permute(0, 2, 1).numpy() - want).max() < 1e-5


def _vote_worker(rank, world, port, out):
    """Whole-scene inference sharded over ranks: each rank votes on its sub-batches, ONE all-reduce of
    the int32 pool per scene.  The pool arithmetic on CPU is the oracle's add_vote (test infrastructure: the product's
    add() is the HIP kernel); what is under test is the sharding and the collective of scene.VotePool."""
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khairil_tum_facade_semantic_segmentation_amd import scene
    from oracle import scene_oracle
    rs = np.random.RandomState(5)                          # identical scene on every rank
    blocks, bp, P, C, bs = 11, 64, 300, 5, 3
    point_idx = rs.randint(0, P, size=(blocks, bp))
    pred = rs.randint(0, C, size=(blocks, bp))
    weight = rs.choice([0.0, 1.0, 2.5, np.inf], size=(blocks, bp))
    pool = scene.VotePool(P, C, torch.device("cpu"))
    mine = scene.shard_batches(blocks, bs, rank, world)
    local = np.zeros((P, C), np.int64)
    for s in mine:
        scene_oracle.add_vote(local, point_idx[s:s + bs], pred[s:s + bs], weight[s:s + bs])
    pool.pool += torch.from_numpy(local).to(torch.int32)
    pool.all_reduce()
    want = np.zeros((P, C), np.int64)
    scene_oracle.add_vote(want, point_idx, pred, weight)
    covered = sorted(s for r in range(world) for s in scene.shard_batches(blocks, bs, r, world))
    ok = covered == list(range(0, blocks, bs)) and np.array_equal(pool.pool.numpy(), want)
    ok = ok and np.array_equal(pool.labels().numpy(), want.argmax(1))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_scene_vote_pool_sharded_world2():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_vote_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}

"""N>1 path on CPU: world_size-2 gloo jobs exercise the product's data-parallel plumbing (ams_amd/dist.py) and pin
the semantics the HIP engine implements with it: SyncBN sums + a global loss normaliser + one gradient all-reduce
reproduce the single-process full-batch step.  The oracle stands in for the HIP kernels here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.dist import ArenaAllReduce, init_from_env, shard_bounds

CI = [0, 1, 2, 10, 11, 13]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    port = _free_port()
    mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)


def _env(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    return init_from_env("gloo")


def _worker_arena(rank, world, port):
    r, w, _ = _env(rank, world, port)
    assert (r, w) == (rank, world)
    arena = torch.zeros(4096, dtype=torch.uint8)
    arena[256:256 + 40].view(torch.float32).copy_(torch.arange(10, dtype=torch.float32) * (rank + 1))
    arena[1024:1024 + 48].view(torch.float64).copy_(torch.full((6,), 0.5 + rank, dtype=torch.float64))
    red = ArenaAllReduce(arena)
    # the engine calls back with (user, byte offset, element count, dtype code)
    assert red(None, 256, 10, hip.DT_F32) == 0
    assert red(None, 1024, 6, hip.DT_F64) == 0
    assert torch.equal(arena[256:296].view(torch.float32), torch.arange(10, dtype=torch.float32) * 3)
    assert torch.equal(arena[1024:1072].view(torch.float64), torch.full((6,), 2.0, dtype=torch.float64))
    assert red.calls == 2 and red.bytes == 40 + 48
    assert red(None, 0, 4, hip.DT_U8) == 1 and isinstance(red.error, ValueError)      # unknown dtype -> error code, no raise
    dist.destroy_process_group()


def test_arena_allreduce_world2():
    _spawn(_worker_arena, 2)


def test_shard_bounds_cover_everything():
    for n in (8, 10, 3, 1):
        for world in (1, 2, 4, 8):
            parts = [shard_bounds(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker_syncbn(rank, world, port, tmp):
    import torch.distributed.nn.functional as dfn
    from oracle.student_torch import StudentOracle
    _env(rank, world, port)
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames, labels = synth.SyntheticVideo(32, 4, CI, seed=5).clip()
    b, e = shard_bounds(len(frames), rank, world)

    class SyncBNOracle(StudentOracle):
        """Same graph; BN statistics and the loss normaliser come from cross-rank sums (what the engine's callback does)."""

        def _bn(self, x, layer, mode, p):
            assert mode == "train"
            g = p[layer.scope + "/BatchNorm/gamma:0"].view(1, -1, 1, 1)
            bta = p[layer.scope + "/BatchNorm/beta:0"].view(1, -1, 1, 1)
            n_local = x.shape[0] * x.shape[2] * x.shape[3]
            s1 = dfn.all_reduce(x.sum(dim=(0, 2, 3)))
            s2 = dfn.all_reduce((x * x).sum(dim=(0, 2, 3)))
            n = n_local * world
            mu = (s1 / n).view(1, -1, 1, 1)
            var = (s2 / n).view(1, -1, 1, 1) - mu * mu
            return (x - mu) * torch.rsqrt(var + layer.bn_eps) * g + bta

    o = SyncBNOracle(W0, CI, dtype=torch.float64)
    params = dict(o.vars)
    leaves = {v.name: o.vars[v.name].clone().requires_grad_(True) for v in o.spec.trainable}
    params.update(leaves)
    z = o.reduced_logits(o.logits_full(frames[b:e].astype(np.float32), "train", params))
    target, weight = o.label_targets(labels[b:e])
    pixel = torch.logsumexp(z, -1) - torch.gather(z, -1, target.unsqueeze(-1)).squeeze(-1)
    valid = weight > 0
    stats = torch.stack([pixel[valid].sum().detach(), valid.sum().to(torch.float64)])
    dist.all_reduce(stats)                                            # global CE sum and valid-pixel count
    loss_local = pixel[valid].sum() / stats[1]
    grads = torch.autograd.grad(loss_local, list(leaves.values()))
    # one flat gradient buffer per rank, summed through the product's callback object
    flat = torch.cat([g.reshape(-1) for g in grads]).to(torch.float32)
    arena = torch.zeros(flat.numel() * 4 + 256, dtype=torch.uint8)
    arena[256:].view(torch.float32).copy_(flat)
    red = ArenaAllReduce(arena)
    assert red(None, 256, flat.numel(), hip.DT_F32) == 0
    if rank == 0:
        np.save(os.path.join(tmp, "dp_grads.npy"), arena[256:].view(torch.float32).numpy())
        np.save(os.path.join(tmp, "dp_loss.npy"), np.array([float(stats[0] / stats[1])]))
    dist.destroy_process_group()


def test_syncbn_dp_equals_full_batch(tmp_path):
    """2 ranks x 2 frames with cross-rank BN sums, a global loss denominator and a summed gradient == 1 process x 4 frames."""
    from oracle.student_torch import StudentOracle
    _spawn(_worker_syncbn, 2, str(tmp_path))
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames, labels = synth.SyntheticVideo(32, 4, CI, seed=5).clip()
    o = StudentOracle(W0, CI, dtype=torch.float64)
    loss, grads = o.gradients(frames.astype(np.float32), labels)
    want = np.concatenate([grads[v.name].numpy().reshape(-1) for v in o.spec.trainable])
    got = np.load(tmp_path / "dp_grads.npy").astype(np.float64)
    assert float(np.load(tmp_path / "dp_loss.npy")[0]) == pytest.approx(loss, rel=1e-9)
    cos = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
    assert cos > 0.999999 and np.abs(got - want).max() / np.abs(want).max() < 1e-5

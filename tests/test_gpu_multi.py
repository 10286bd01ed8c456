"""Multi-GPU: the batch-sharded fine-tune step over the library's own RCCL communicator, one process per GPU (the reference runs one process
per --gpu: run.py:28-29, SemanticNetwork.py:74; BASELINE.json configs[4]).  Needs >= 2 visible GPUs: on a one-GPU box these tests SKIP —
they exist so that the first multi-GPU box gives a correctness signal and not only a timing."""
import os
import socket

import numpy as np
import pytest
import torch

from ams_amd import spec as S, synth, weights as Wt

pytestmark = pytest.mark.gpu

CI = [0, 1, 2, 10, 11, 13]
H = 64
PER_RANK = 2


def _rccl_worker(rank, world, port, tmp):
    """A fresh process (spawn): its first GPU call is set_device(rank).  torch.distributed (gloo) only carries the 128-byte RCCL id."""
    import torch.distributed as dist
    from ams_amd.dist import RcclComm, init_from_env, shard_bounds
    from ams_amd.engine import StudentEngine
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    init_from_env("gloo")
    dev = torch.device("cuda", rank)
    comm = RcclComm(rank, world, device=dev)
    assert comm.rank_world() == (rank, world), comm.rank_world()          # RCCL's own view of the job, not the launcher's environment
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    n = PER_RANK * world
    frames, labels = synth.SyntheticVideo(H, n, CI, seed=9).clip()
    b, e = shard_bounds(n, rank, world)
    eng = StudentEngine(CI, H, 2 * H, max_batch=PER_RANK, trainable=True, device=dev)
    eng.load_variables(W0)
    # a plain all-reduce first: every rank contributes rank + 1
    t = torch.full((1000,), float(rank + 1), device=dev)
    comm.all_reduce(t)
    torch.cuda.synchronize(dev)
    assert torch.all(t == world * (world + 1) / 2), t[:4]
    calls0, _ = comm.stats()
    ls = eng.train_step(frames[b:e], labels[b:e], 1e-3, comm=comm, global_batch=n).cpu().numpy()
    torch.cuda.synchronize(dev)
    calls, nbytes = comm.stats()
    np.save(os.path.join(tmp, "rccl_params_%d.npy" % rank), eng.params.cpu().numpy())
    if rank == 0:
        np.save(os.path.join(tmp, "rccl_grads.npy"), eng.grads.cpu().numpy())
        np.save(os.path.join(tmp, "rccl_stats.npy"), eng.stats.cpu().numpy())
        np.save(os.path.join(tmp, "rccl_loss.npy"), ls)
        np.save(os.path.join(tmp, "rccl_calls.npy"), np.array([calls - calls0, nbytes]))
    eng.close()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_batch_sharded_step_equals_single_process(world, tmp_path):
    """`world` processes, one GPU each, one fine-tune step on a shard of the batch each through RcclComm (every cross-rank sum one ncclAllReduce
    on the launch stream) == the single-process step on the whole batch: global loss and valid-pixel count, gradients, SyncBN moving
    averages, identical parameters on every rank, 110 collectives (54 BN forward + loss + 54 BN backward + the gradient arena)."""
    n_gpus = torch.cuda.device_count()           # (counting devices does not initialise the GPU)
    if n_gpus < world:
        pytest.skip("multi-rank RCCL needs %d GPUs; this box shows %d" % (world, n_gpus))
    import torch.multiprocessing as mp
    from ams_amd.engine import StudentEngine
    mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n = PER_RANK * world
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames, labels = synth.SyntheticVideo(H, n, CI, seed=9).clip()
    eng = StudentEngine(CI, H, 2 * H, max_batch=n, trainable=True)
    eng.load_variables(W0)
    ls = eng.train_step(frames, labels, 1e-3).cpu().numpy()
    dp_ls = np.load(tmp_path / "rccl_loss.npy")
    # the bars of test_data_parallel_step_equals_single_process (gloo, one shared GPU): the count is exact, the CE sum carries the f32
    # summation order of the ranks' BN statistics
    assert dp_ls[1] == ls[1] and dp_ls[0] == pytest.approx(ls[0], rel=1e-5)
    g, dg = eng.grads.cpu().numpy().astype(np.float64), np.load(tmp_path / "rccl_grads.npy").astype(np.float64)
    cos = float(g @ dg / (np.linalg.norm(g) * np.linalg.norm(dg)))
    assert cos > 0.99999, cos
    st, dst = eng.stats.cpu().numpy().astype(np.float64), np.load(tmp_path / "rccl_stats.npy").astype(np.float64)
    assert np.abs(st - dst).max() / np.abs(st).max() < 1e-5
    p0 = np.load(tmp_path / "rccl_params_0.npy")
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / ("rccl_params_%d.npy" % r)), p0), "rank %d holds other parameters than rank 0" % r
    calls, nbytes = np.load(tmp_path / "rccl_calls.npy")
    assert calls == 54 * 2 + 2
    assert nbytes > 4 * eng.spec.n_trainable
    eng.close()

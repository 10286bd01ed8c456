"""create_student_v3's remaining kwargs on the HIP path (utils/graph_utils.py:338-339): soft_teacher — the fine-tune loss against cached teacher
LOGITS (:375-376, 403-408; BASELINE.json north_star "KD loss against cached teacher logits") — and regularize / train_biases_only (:451-456).
Kernel level against f64 autograd, step level against the f64 oracle (small size and one 512 x 1024 step), the SemanticNetwork surface."""
import ctypes as C
import random
from collections import deque

import numpy as np
import pytest
import torch

from ams_amd import exp_configs, hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from ams_amd.semantic_network import SemanticNetwork

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CI = [0, 1, 2, 10, 11, 13]


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


@pytest.fixture(scope="module")
def W0():
    return Wt.synthetic_weights(S.build_spec(), seed=0)


def _teacher_logits(labels, rng, nc=19, th=None, tw=None, sharp=3.0):
    """Synthetic cached teacher logits: noise + a bump on the class of the label map (what a teacher whose argmax gave those labels looks like)."""
    B, H, W = labels.shape
    t = rng.standard_normal((B, H, W, nc)).astype(np.float32)
    ok = labels < nc
    bi, yi, xi = np.nonzero(ok)
    t[bi, yi, xi, labels[ok]] += sharp
    if th is not None:
        ys = np.round(np.linspace(0, H - 1, th)).astype(int)
        xs = np.round(np.linspace(0, W - 1, tw)).astype(int)
        t = np.ascontiguousarray(t[:, ys][:, :, xs])
    return t


@pytest.mark.parametrize("h,w,H,W,cls,th,tw", [(5, 9, 64, 128, CI, 64, 128), (3, 5, 32, 64, [2, 8, 9, 10, 11, 13], 32, 64),
                                               (9, 17, 128, 256, list(range(19)), 128, 256), (4, 7, 50, 90, [0, 15], 50, 90),
                                               (5, 9, 64, 128, CI, 5, 9), (4, 7, 50, 90, [0, 15], 13, 31), (5, 9, 64, 128, CI, 1, 1)])
def test_soft_teacher_loss_and_gradient_kernel(h, w, H, W, cls, th, tw):
    """ams_k_ce_loss_grad_soft against f64 autograd of -sum_k softmax(gather(t))_k log softmax(gather(resize(z)))_k, masked by the hard labels:
    teacher logits at the label size (the reference's feed) and on smaller grids (resized like the student's logits); run-to-run identical."""
    from oracle.student_torch import resize_bilinear_align_corners
    lib = hip.lib()
    rng = np.random.default_rng(h * w + th)
    B, NC, K = 2, 19, len(cls)
    logits = (rng.standard_normal((B, h, w, NC)) * 2).astype(np.float32)
    teacher = rng.integers(0, 19, (B, H, W)).astype(np.uint8)
    teacher[rng.random((B, H, W)) < 0.1] = 255
    tl = _teacher_logits(np.where(teacher < 19, teacher, 0).astype(np.int64), rng, th=th if (th, tw) != (H, W) else None, tw=tw)
    assert tl.shape == (B, th, tw, NC)
    ci = (C.c_int32 * K)(*cls)
    ld, td, tld = torch.as_tensor(logits).to(DEV), torch.as_tensor(teacher).to(DEV), torch.as_tensor(tl).to(DEV)
    lut = np.full(256, -1)
    lut[cls] = np.arange(K)
    valid = torch.as_tensor(lut[teacher] >= 0)
    lt = torch.as_tensor(logits).double().requires_grad_(True)
    zf = resize_bilinear_align_corners(lt, H, W)[..., cls]
    tf_ = torch.as_tensor(tl).double()
    if (th, tw) != (H, W):
        tf_ = resize_bilinear_align_corners(tf_, H, W)
    p = torch.softmax(tf_[..., cls], -1)
    pix = -(p * torch.log_softmax(zf, -1)).sum(-1)
    want = pix[valid].mean()
    want.backward()
    n = lib.ams_k_ce_loss_grad_scratch(B, h, w, K)
    scr = torch.full((n,), np.nan, device=DEV)
    outs = []
    for _ in range(2):
        loss = torch.full((2,), np.nan, dtype=torch.float64, device=DEV)
        dl = torch.full((B, h, w, NC), np.nan, device=DEV)
        hip.check(lib.ams_k_ce_loss_grad_soft(P(ld), B, h, w, NC, ci, K, H, W, P(td), P(tld), th, tw, P(loss), P(dl), P(scr), n, stream()))
        outs.append((loss.cpu().numpy(), dl.cpu().numpy()))
    got, d = outs[0]
    assert got[1] == int(valid.sum())
    assert got[0] / got[1] == pytest.approx(float(want), rel=2e-5)
    assert rel_err(d, lt.grad.numpy()) < 3e-5
    unsel = [c for c in range(NC) if c not in cls]
    assert np.all(d[..., unsel] == 0)
    assert np.array_equal(d, outs[1][1]) and np.array_equal(got, outs[1][0])
    # one-hot limit: all of the teacher's mass on the label's class = the hard-label kernel's loss and gradient
    peaked = np.full((B, H, W, NC), -1e4, dtype=np.float32)
    bi, yi, xi = np.nonzero(teacher < 19)
    peaked[bi, yi, xi, teacher[teacher < 19]] = 1e4
    pk = torch.as_tensor(peaked).to(DEV)
    loss_s = torch.empty(2, dtype=torch.float64, device=DEV)
    dl_s = torch.empty((B, h, w, NC), device=DEV)
    hip.check(lib.ams_k_ce_loss_grad_soft(P(ld), B, h, w, NC, ci, K, H, W, P(td), P(pk), H, W, P(loss_s), P(dl_s), P(scr), n, stream()))
    loss_h = torch.empty(2, dtype=torch.float64, device=DEV)
    dl_h = torch.empty((B, h, w, NC), device=DEV)
    hip.check(lib.ams_k_ce_loss_grad(P(ld), B, h, w, NC, ci, K, H, W, P(td), P(loss_h), P(dl_h), P(scr), n, stream()))
    assert loss_s.cpu().numpy()[0] == pytest.approx(loss_h.cpu().numpy()[0], rel=1e-6)
    assert rel_err(dl_s.cpu().numpy(), dl_h.cpu().numpy()) < 1e-6


def _grad_report(eng, grads_o):
    g = eng.grads.cpu().numpy().astype(np.float64)
    flat = np.concatenate([grads_o[v.name].numpy().reshape(-1) for v in eng.spec.trainable])
    cos = float(g @ flat / (np.linalg.norm(g) * np.linalg.norm(flat)))
    return g, flat, cos


@pytest.mark.parametrize("low_res", [False, True])
def test_soft_teacher_step_matches_f64_oracle(W0, low_res):
    """One fine-tune step of the soft_teacher graph at 64 x 128 against the f64 oracle: loss to 1e-3, the gradient in the f32 error class
    (cosine; the worst tensors bounded as in test_train_step_matches_oracle), and it is NOT the hard-label step."""
    from oracle.student_torch import StudentOracle
    H, B, lr = 64, 4, 1e-3
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=5).clip()
    rng = np.random.default_rng(17)
    tl = _teacher_logits(labels.astype(np.int64), rng, th=9 if low_res else None, tw=17)
    o = StudentOracle(W0, CI, dtype=torch.float64)
    o32 = StudentOracle(W0, CI)
    loss_o, grads_o = o.gradients(frames.astype(np.float32), labels, teacher_logits=tl)
    _, grads_32 = o32.gradients(frames.astype(np.float32), labels, teacher_logits=tl)
    loss_hard, _ = o.gradients(frames.astype(np.float32), labels)
    assert abs(loss_o - loss_hard) > 1e-2 * abs(loss_hard)
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W0)
    eng.set_soft_teacher(True)
    with pytest.raises(AssertionError):
        eng.train_step(frames, labels, lr)                                       # teacher_labels_logits_pl not fed
    ls = eng.train_step(frames, labels, lr, teacher_logits=tl).cpu().numpy()
    assert ls[0] / ls[1] == pytest.approx(loss_o, rel=1e-3)
    assert ls[1] == float(np.isin(labels, CI).sum())
    g, flat, cos = _grad_report(eng, grads_o)
    assert cos > 0.9995, cos
    gnorm = max(float(gv.abs().max()) for gv in grads_o.values())
    for v in eng.spec.trainable:
        want = grads_o[v.name].numpy().reshape(-1)
        floor = max(np.linalg.norm(want), 1e-3 * gnorm * np.sqrt(want.size))
        e_gpu = np.linalg.norm(g[v.offset:v.offset + v.size] - want) / floor
        e_f32 = np.linalg.norm(grads_32[v.name].numpy().reshape(-1).astype(np.float64) - want) / floor
        assert e_gpu < max(6e-2, 4 * e_f32), (v.name, e_gpu, e_f32)
    # the C ABI refuses a soft step whose feed was cleared (TensorFlow: "You must feed a value for placeholder tensor")
    hip.check(eng.lib.ams_student_feed_teacher_logits(eng._h, None, 0, 0))
    t, dt, b = eng._frames_to_device(frames)
    lab = eng._labels_to_device(labels, b)
    rc = eng.lib.ams_student_train_step(eng._h, C.c_void_p(t.data_ptr()), dt, C.c_void_p(lab.data_ptr()), b, lr, None, None,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"teacher logits" in eng.lib.ams_last_error()
    # back to the hard-label graph
    eng.load_variables(W0)
    eng.set_soft_teacher(False)
    ls_h = eng.train_step(frames, labels, lr).cpu().numpy()
    assert ls_h[0] / ls_h[1] == pytest.approx(loss_hard, rel=1e-3)
    eng.close()


def test_soft_teacher_step_full_size(W0):
    """One 2-frame soft-teacher step at 512 x 1024 (BASELINE.json configs[2]'s size) with teacher logits at the label size — the reference's
    feed, 80 MB per step — and with the same logits cached at output stride 16: against the f64 oracle."""
    from oracle.student_torch import StudentOracle
    H, B = 512, 2
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=1).clip()
    rng = np.random.default_rng(3)
    o = StudentOracle(W0, CI, dtype=torch.float64)
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.set_soft_teacher(True)
    for th, tw in ((None, None), (33, 65)):
        tl = _teacher_logits(labels.astype(np.int64), rng, th=th, tw=tw)
        loss_o, grads_o = o.gradients(frames.astype(np.float32), labels, teacher_logits=tl)
        eng.load_variables(W0)
        ls = eng.train_step(frames, labels, 1e-3, teacher_logits=tl).cpu().numpy()
        assert ls[0] / ls[1] == pytest.approx(loss_o, rel=1e-3), (th, tw)
        g, flat, cos = _grad_report(eng, grads_o)
        assert cos > 0.9995, (th, tw, cos)
        for name in ("aspp0/weights:0", "MobilenetV2/expanded_conv_16/project/weights:0", "logits/semantic/weights:0"):
            v = eng.spec.by_name[name]
            want = grads_o[name].numpy().reshape(-1)
            e = np.linalg.norm(g[v.offset:v.offset + v.size] - want) / np.linalg.norm(want)
            assert e < 3e-2, (name, e)
    eng.close()


@pytest.mark.parametrize("biases_only", [False, True])
def test_regularized_step_matches_f64_oracle(W0, biases_only):
    """regularize=True (and train_biases_only): the reported loss carries 0.01 * mean l2 and every regularised tensor's gradient moves by
    (0.01 / n_vars) v — exactly the difference of the oracle's two gradients — while the others keep the plain step's bits."""
    from oracle.student_torch import StudentOracle
    H, B, lr = 64, 2, 1e-3
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=9).clip()
    o = StudentOracle(W0, CI, dtype=torch.float64)
    loss_p, grads_p = o.gradients(frames.astype(np.float32), labels)
    loss_r, grads_r = o.gradients(frames.astype(np.float32), labels, regularize=True, train_biases_only=biases_only)
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W0)
    ls_p = eng.train_step(frames, labels, lr).cpu().numpy()
    g_plain = eng.grads.cpu().numpy().copy()
    eng.load_variables(W0)
    eng.set_regularizer(True, biases_only=biases_only)
    ls_r = eng.train_step(frames, labels, lr).cpu().numpy()
    g_reg = eng.grads.cpu().numpy()
    assert ls_r[1] == ls_p[1]
    assert ls_r[0] / ls_r[1] - ls_p[0] / ls_p[1] == pytest.approx(loss_r - loss_p, rel=1e-5)
    n_reg = 0
    for v in eng.spec.trainable:
        sl = slice(v.offset, v.offset + v.size)
        want = (grads_r[v.name] - grads_p[v.name]).numpy().reshape(-1)
        if biases_only and 'weight' in v.name:
            assert not want.any() and np.array_equal(g_reg[sl], g_plain[sl]), v.name
        else:
            n_reg += 1
            got = g_reg[sl].astype(np.float64) - g_plain[sl]
            assert np.abs(got - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-12) + 2e-7 * np.abs(g_plain[sl]).max(), v.name
    assert n_reg == (109 if biases_only else 164)
    eng.set_regularizer(False)
    eng.load_variables(W0)
    eng.train_step(frames, labels, lr)
    assert np.array_equal(eng.grads.cpu().numpy(), g_plain)
    eng.close()


def test_semantic_network_soft_teacher_surface(W0):
    """SemanticNetwork(soft_teacher=True, regularize=True): the constructor kwargs of the reference (SemanticNetwork.py:140-152 forwards them to
    create_student_v3), train_step with the batch's logits, train_with_deque with the replay memory's cached logits — the batches a seeded run
    draws take the logits of the frames they drew (same losses as explicit steps on those picks)."""
    H = 64
    frames, labels = synth.SyntheticVideo(H, 6, CI, seed=4).clip()
    rng = np.random.default_rng(8)
    tl = _teacher_logits(labels.astype(np.int64), rng, th=5, tw=9)
    kw = dict(class_weights_exp=exp_configs.class_weights(25), height=H, scale=[1], mini_batch_size=2, lr=1e-3, initial_variables=W0)
    net = SemanticNetwork("unused", soft_teacher=True, **kw)
    with pytest.raises(AssertionError):
        net.train_with_deque(deque(frames), deque(labels), 1)                    # soft graph without its feed
    with pytest.raises(AssertionError):
        net.train_step(frames[:2], labels[:2])
    np.random.seed(11)
    random.seed(11)
    net.train_with_deque(deque(frames), deque(labels), 3, teacher_logits_deque=deque(tl))
    got = list(net.last_losses)
    assert len(got) == 3 and all(np.isfinite(got))
    # the same picks, fed explicitly
    ref = SemanticNetwork("unused", soft_teacher=True, **kw)
    np.random.seed(11)
    random.seed(11)
    want = []
    for _ in range(3):
        picks = []
        for _j in range(2):
            picks.append(np.random.choice(len(frames)))
            random.randint(0, 0); random.randint(0, 0); random.randint(0, 0)
        want.append(ref.train_step(frames[picks], labels[picks], teacher_logits=tl[picks]))
    assert got == pytest.approx(want, rel=1e-6)
    a, b = net.get_vars(), ref.get_vars()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    hard = SemanticNetwork("unused", **kw)
    with pytest.raises(AssertionError):
        hard.train_step(frames[:2], labels[:2], teacher_logits=tl[:2])           # logits without the soft graph
    lh = hard.train_step(frames[:2], labels[:2])
    reg = SemanticNetwork("unused", regularize=True, train_biases_only=True, **kw)
    lr_ = reg.train_step(frames[:2], labels[:2])
    assert lr_ > lh and lr_ - lh == pytest.approx(0.01 * np.mean([float((np.asarray(W0[v.name], np.float64) ** 2).sum() / 2)
                                                                    for v in reg.engine.spec.trainable if 'weight' not in v.name]), rel=1e-4)
    for n in (net, ref, hard, reg):
        n.close_model()


# ---------------------------------------------------------------------------------------------------------
# the soft-teacher step sharded over the batch: 2 ranks (gloo, both on cuda:0) x 2 frames == 1 rank x 4 frames
# ---------------------------------------------------------------------------------------------------------
def _dp_soft_worker(rank, world, port, tmp):
    import os
    import torch.distributed as dist
    from ams_amd.dist import ArenaAllReduce, init_from_env, shard_bounds
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_from_env("gloo")
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames, labels = synth.SyntheticVideo(64, 4, CI, seed=9).clip()
    tl = _teacher_logits(labels.astype(np.int64), np.random.default_rng(21), th=9, tw=17)
    b, e = shard_bounds(4, rank, world)
    eng = StudentEngine(CI, 64, 128, max_batch=2, trainable=True)
    eng.load_variables(W0)
    eng.set_soft_teacher(True)
    red = ArenaAllReduce(eng.arena)
    ls = eng.train_step(frames[b:e], labels[b:e], 1e-3, allreduce=red, global_batch=4, teacher_logits=tl[b:e]).cpu().numpy()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(tmp, "dps_grads.npy"), eng.grads.cpu().numpy())
        np.save(os.path.join(tmp, "dps_loss.npy"), ls)
    eng.close()
    dist.destroy_process_group()


def test_data_parallel_soft_teacher_step_equals_single_process(W0, tmp_path):
    """Every rank feeds the cached teacher logits of ITS shard; the loss sums and the valid-pixel count are all-reduced as in the hard-label step:
    the sharded soft step is the single-process step on the whole batch."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_dp_soft_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    frames, labels = synth.SyntheticVideo(64, 4, CI, seed=9).clip()
    tl = _teacher_logits(labels.astype(np.int64), np.random.default_rng(21), th=9, tw=17)
    eng = StudentEngine(CI, 64, 128, max_batch=4, trainable=True)
    eng.load_variables(W0)
    eng.set_soft_teacher(True)
    ls = eng.train_step(frames, labels, 1e-3, teacher_logits=tl).cpu().numpy()
    dp_ls = np.load(tmp_path / "dps_loss.npy")
    assert dp_ls[1] == ls[1] and dp_ls[0] == pytest.approx(ls[0], rel=1e-5)
    g, dg = eng.grads.cpu().numpy().astype(np.float64), np.load(tmp_path / "dps_grads.npy").astype(np.float64)
    cos = float(g @ dg / (np.linalg.norm(g) * np.linalg.norm(dg)))
    assert cos > 0.99999, cos
    eng.close()

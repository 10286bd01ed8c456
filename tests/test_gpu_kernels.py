"""Kernel-level parity: every HIP kernel, called through the C ABI, against the CPU oracle's op of the same name.

Tolerances: f32 kernels vs an f64 evaluation of the same math — relative error <= 2e-5 of the output scale for
GEMM-like reductions (f32 round-off grows ~sqrt(K)), exact equality for integer outputs (labels, confusion
matrix).  Sizes are ragged on purpose (M not a multiple of the tile, odd spatial sizes, K=24, N=19).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ams_amd import hip, spec as S

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip.lib()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(DEV).contiguous()


_KEEP = []


def PD(a, dtype=torch.float32):
    """device copy kept alive until the end of the test session (a bare PD(a) frees the tensor at once)."""
    t = dev(a, dtype)
    _KEEP.append(t)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return P(t)


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------ pointwise
@pytest.mark.parametrize("M,K,N", [(1000, 16, 96), (4097, 96, 24), (513, 24, 144), (2145, 960, 320), (2145, 160, 960),
                                   (300, 320, 256), (77, 256, 19), (3, 320, 256), (70000, 32, 16), (2145, 576, 160),
                                   (129 * 257, 144, 32), (2145, 384, 64)])
def test_pointwise_forward(lib, M, K, N):
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32)
    shift = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    xd, wd, sd, hd, rd = dev(x), dev(w), dev(scale), dev(shift), dev(res)
    y = torch.empty((M, N), device=DEV)
    hip.check(lib.ams_k_pointwise(P(xd), M, K, P(wd), N, 0, None, 1, P(sd), P(hd), hip.ACT_RELU6, P(rd), P(y), stream()))
    want = np.clip((x.astype(np.float64) @ w.astype(np.float64)) * scale + shift, 0, 6) + res
    assert rel_err(y.cpu().numpy(), want) < 2e-5
    # plain product, no epilogue
    hip.check(lib.ams_k_pointwise(P(xd), M, K, P(wd), N, 0, None, 1, None, None, hip.ACT_NONE, None, P(y), stream()))
    assert rel_err(y.cpu().numpy(), x.astype(np.float64) @ w.astype(np.float64)) < 2e-5


def test_pointwise_image_bias_and_transposed_weights(lib):
    rng = np.random.default_rng(5)
    B, HW, K, N = 3, 715, 256, 256
    M = B * HW
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / 16).astype(np.float32)
    bias = rng.standard_normal((B, N)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32)
    shift = rng.standard_normal(N).astype(np.float32)
    y = torch.empty((M, N), device=DEV)
    hip.check(lib.ams_k_pointwise(PD(x), M, K, PD(w), N, 0, PD(bias), HW, PD(scale), PD(shift),
                                  hip.ACT_RELU, None, P(y), stream()))
    want = np.maximum((x.astype(np.float64) @ w + np.repeat(bias, HW, axis=0)) * scale + shift, 0)
    assert rel_err(y.cpu().numpy(), want) < 2e-5
    # dgrad form: y = x @ wt^T with wt stored [N_out, K_in] = the forward weight [K_fwd, N_fwd]
    wt = (rng.standard_normal((96, K)) / 16).astype(np.float32)          # forward weight [K_fwd=96, N_fwd=256]
    y2 = torch.empty((M, 96), device=DEV)
    hip.check(lib.ams_k_pointwise(PD(x), M, K, PD(wt), 96, 1, None, 1, None, None, hip.ACT_NONE, None, P(y2), stream()))
    assert rel_err(y2.cpu().numpy(), x.astype(np.float64) @ wt.T.astype(np.float64)) < 2e-5


@pytest.mark.parametrize("M,K,N", [(2145, 960, 320), (17160, 160, 960), (2145, 384, 64), (4290, 576, 160), (2145, 256, 19),
                                   (1000, 64, 384), (300, 320, 256), (2145, 24 * 8, 32)])
def test_pointwise_split_bf16(lib, M, K, N):
    """bf16x3 product: error budget 2^-16 per product -> ~1e-5 of the output scale; must beat plain bf16 (4e-3) by far."""
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32)
    shift = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    Kp = (K + 31) // 32 * 32
    panels = torch.zeros(2 * N * Kp, dtype=torch.int16, device=DEV)
    y = torch.empty((M, N), device=DEV)
    hip.check(lib.ams_k_pointwise_split(PD(x), M, K, PD(w), N, PD(scale), PD(shift), hip.ACT_RELU6, PD(res), P(y), P(panels),
                                        panels.numel(), stream()))
    want = np.clip((x.astype(np.float64) @ w.astype(np.float64)) * scale + shift, 0, 6) + res
    assert rel_err(y.cpu().numpy(), want) < 5e-5
    hip.check(lib.ams_k_pointwise_split(PD(x), M, K, PD(w), N, None, None, hip.ACT_NONE, None, P(y), P(panels),
                                        panels.numel(), stream()))
    assert rel_err(y.cpu().numpy(), x.astype(np.float64) @ w.astype(np.float64)) < 5e-5


@pytest.mark.parametrize("M,K,N", [(5000, 16, 96), (4097, 96, 24), (2145 * 2, 960, 320), (2145, 160, 960), (3, 320, 256),
                                   (33000, 32, 16), (2145, 256, 19), (9000, 27, 32), (70000, 144, 24)])
def test_pointwise_wgrad(lib, M, K, N):
    rng = np.random.default_rng(M * 3 + K + N)
    ldx = 32 if K == 27 else K
    x = rng.standard_normal((M, ldx)).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    n_scr = lib.ams_k_pointwise_wgrad_scratch(M, K, N)
    scr = torch.empty(n_scr, device=DEV)
    dw = torch.full((K, N), np.nan, device=DEV)
    if ldx != K:
        pytest.skip("padded leading dimension is exercised through the engine's stem path")
    hip.check(lib.ams_k_pointwise_wgrad(PD(x), PD(dy), M, K, N, P(dw), P(scr), n_scr, stream()))
    want = x[:, :K].astype(np.float64).T @ dy.astype(np.float64)
    assert rel_err(dw.cpu().numpy(), want) < 3e-5
    # deterministic: a second run gives the same bits
    dw2 = torch.empty((K, N), device=DEV)
    hip.check(lib.ams_k_pointwise_wgrad(PD(x), PD(dy), M, K, N, P(dw2), P(scr), n_scr, stream()))
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("M,K,N", [(16384, 960, 160), (2048, 64, 384), (4160, 384, 64), (1500, 576, 96), (1024, 320, 256),
                                   (3000, 144, 24 * 4), (2080, 256, 20)])
def test_pointwise_wgrad_split(lib, M, K, N):
    """3-part bf16 split on the matrix pipe (transpose reads from LDS): f32-level accuracy, deterministic, ragged M / N."""
    rng = np.random.default_rng(M + K + N)
    x = (rng.standard_normal((M, K)) * rng.uniform(0.1, 4.0, (1, K))).astype(np.float32)
    dy = (rng.standard_normal((M, N)) * 1e-3).astype(np.float32)
    n_scr = lib.ams_k_pointwise_wgrad_scratch(M, K, N)
    scr = torch.empty(n_scr, device=DEV)
    dw = torch.full((K, N), np.nan, device=DEV)
    hip.check(lib.ams_k_pointwise_wgrad_split(PD(x), PD(dy), M, K, N, P(dw), P(scr), n_scr, stream()))
    want = x.astype(np.float64).T @ dy.astype(np.float64)
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    # against the exact-f32 MFMA kernel's own bar (3e-5): the split must be in the same class
    assert rel_err(got, want) < 3e-5
    f32 = torch.empty((K, N), device=DEV)
    hip.check(lib.ams_k_pointwise_wgrad(PD(x), PD(dy), M, K, N, P(f32), P(scr), n_scr, stream()))
    assert rel_err(got, want) <= 2 * rel_err(f32.cpu().numpy(), want) + 1e-6
    dw2 = torch.empty((K, N), device=DEV)
    hip.check(lib.ams_k_pointwise_wgrad_split(PD(x), PD(dy), M, K, N, P(dw2), P(scr), n_scr, stream()))
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("M,K,N,split,trans", [(4290, 160, 960, 1, 1), (16421, 960, 160, 1, 0), (17160, 576, 96, 1, 1), (2145, 384, 64, 1, 0),
                                                (33001, 64, 384, 1, 0), (40003, 24, 144, 0, 1), (33000, 32, 192, 0, 0), (65537, 16, 96, 0, 1),
                                                (70001, 144, 24, 0, 0), (200003, 24, 96, 0, 1), (131075, 32, 192, 0, 1),
                                                (17160, 960, 160, 2, 0), (17160, 160, 960, 2, 0), (16421, 576, 96, 2, 0), (2145, 384, 64, 2, 0),
                                                (33001, 64, 384, 2, 0)])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pointwise_with_fused_column_reduction(lib, M, K, N, split, trans, mode):
    """1x1 GEMMs whose epilogue also reduces (PwArgs::red_mode): forward statistics (mode 1), BN-backward sums with the activation's
    derivative applied to the stored result (mode 2), the same with a residual gradient added first (3 = mode 2 + res) — tiled three-part
    kernel and streaming exact-f32 kernel, ragged row counts (tail strips, half-height tail blocks), both weight orientations; the last two
    shapes are the early blocks' input-gradient GEMMs at row counts where every wave of the streaming kernel's co-resident grid walks several row
    groups of its 48-wide column tile."""
    import ctypes as C
    if split == 2 and mode != 1:
        pytest.skip("the two-fp16-part kernel fuses the FORWARD statistics only (gradients leave fp16's range)")
    rng = np.random.default_rng(M + K + N + mode)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K) if trans else (K, N)) / np.sqrt(K)).astype(np.float32)
    center = (rng.standard_normal(N) * 0.1).astype(np.float32)
    z = (rng.standard_normal((M, N)) * 2).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32)
    shift = (rng.standard_normal(N) + 1).astype(np.float32)
    mean = (rng.standard_normal(N) * 0.2).astype(np.float32)
    rstd = rng.uniform(0.5, 2.0, N).astype(np.float32)
    # no activation within rounding distance of a ReLU6 knee (the f32 and f64 masks must agree element for element)
    a0 = z.astype(np.float64) * scale + shift
    near = (np.abs(a0) < 1e-3) | (np.abs(a0 - 6) < 1e-3)
    z = np.where(near, z + np.float32(0.01) / scale, z).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32) if mode == 3 else None
    prod = x.astype(np.float64) @ (w.astype(np.float64).T if trans else w.astype(np.float64))
    y = torch.empty((M, N), device=DEV)
    n_part = max(M // 64 + 8, 2048) * 2 * N
    part = torch.full((n_part,), float("nan"), device=DEV)
    Kp = (K + 31) // 32 * 32
    panels = torch.empty(3 * N * Kp, dtype=torch.int16, device=DEV)
    rows = C.c_int32(-1)
    hip.check(lib.ams_k_pointwise_red(PD(x), M, K, PD(w), N, trans, split, 1 if mode == 1 else 2, PD(center), PD(z), PD(scale), PD(shift), PD(mean),
                                      PD(rstd), hip.ACT_RELU6, PD(res) if res is not None else None, P(y), P(part), n_part, C.byref(rows),
                                      P(panels), panels.numel(), stream()))
    tol = 3e-5 if split else 1e-5
    if mode == 3 and not split:
        # the streaming kernel does not take a residual into its reduction: it must say so and return the plain sum
        assert rows.value == 0
        assert rel_err(y.cpu().numpy(), prod + res) < tol
        return
    assert rows.value > 0, "the kernel chosen for this shape should fuse the reduction"
    sums = part[: rows.value * 2 * N].cpu().numpy().astype(np.float64).reshape(rows.value, 2, N).sum(axis=0)
    if mode == 1:
        assert rel_err(y.cpu().numpy(), prod) < tol
        d = prod - center
        assert rel_err(sums[0], d.sum(axis=0)) < 5e-5 and rel_err(sums[1], (d * d).sum(axis=0)) < 5e-5
    else:
        g = prod + (res if res is not None else 0.0)
        a = z.astype(np.float64) * scale + shift
        dy = g * ((a > 0) & (a < 6))
        assert rel_err(y.cpu().numpy(), dy) < tol
        xhat = (z.astype(np.float64) - mean) * rstd
        assert rel_err(sums[0], dy.sum(axis=0)) < 5e-5 and rel_err(sums[1], (dy * xhat).sum(axis=0)) < 5e-5


def test_pointwise_wgrad_split_rejects_small_problems(lib):
    scr = torch.empty(1024, device=DEV)
    z = torch.zeros(16, device=DEV)
    assert lib.ams_k_pointwise_wgrad_split(P(z), P(z), 64, 16, 16, P(z), P(scr), 1024, stream()) == -1


# ------------------------------------------------------------------------------------------------ stem
@pytest.mark.parametrize("H,W,dtype", [(32, 64, np.uint8), (37, 50, np.float32), (64, 128, np.uint8)])
def test_stem_conv(lib, H, W, dtype):
    rng = np.random.default_rng(H)
    B = 2
    frames = rng.integers(0, 256, (B, H, W, 3)).astype(dtype)
    if dtype == np.float32:
        frames = frames + rng.random((B, H, W, 3)).astype(np.float32) * 0.5
    w = (rng.standard_normal((3, 3, 3, 32)) * 0.3).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    shift = rng.standard_normal(32).astype(np.float32)
    Ho, Wo = S.same_pad(H + 1, 3, 2, 1)[0], S.same_pad(W + 1, 3, 2, 1)[0]
    y = torch.empty((B, Ho, Wo, 32), device=DEV)
    fd = torch.as_tensor(frames).to(DEV)
    hip.check(lib.ams_k_stem_conv(P(fd), hip.DT_U8 if dtype == np.uint8 else hip.DT_F32, B, H, W, PD(w), 32, PD(scale),
                                  PD(shift), hip.ACT_RELU6, S.PIXEL_SCALE, P(y), stream()))
    x = torch.as_tensor(frames.astype(np.float32)).permute(0, 3, 1, 2)
    x = F.pad(x, (0, 1, 0, 1), value=127.5) * np.float32(S.PIXEL_SCALE) - 1.0
    _, pt, pb = S.same_pad(H + 1, 3, 2, 1)
    _, pl, pr = S.same_pad(W + 1, 3, 2, 1)
    ref = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), torch.as_tensor(w).double().permute(3, 2, 0, 1), stride=2)
    ref = torch.clamp(ref * torch.as_tensor(scale).view(1, -1, 1, 1) + torch.as_tensor(shift).view(1, -1, 1, 1), 0, 6)
    assert rel_err(y.cpu().numpy(), ref.permute(0, 2, 3, 1).numpy()) < 1e-5


# ------------------------------------------------------------------------------------------------ depthwise
@pytest.mark.parametrize("H,W,Cn,stride,rate", [(33, 65, 32, 1, 1), (33, 65, 96, 2, 1), (17, 33, 960, 1, 2), (20, 31, 144, 2, 1),
                                                (9, 9, 576, 1, 1), (12, 10, 192, 1, 2), (65, 129, 24, 1, 1), (8, 8, 384, 2, 1)])
def test_depthwise_forward_and_backward(lib, H, W, Cn, stride, rate):
    rng = np.random.default_rng(H * W + Cn)
    B = 2
    x = rng.standard_normal((B, H, W, Cn)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cn, 1)) * 0.4).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cn).astype(np.float32)
    shift = rng.standard_normal(Cn).astype(np.float32)
    Ho, pt, pb = S.same_pad(H, 3, stride, rate)
    Wo, pl, pr = S.same_pad(W, 3, stride, rate)
    xd, wd = dev(x), dev(w)
    y = torch.empty((B, Ho, Wo, Cn), device=DEV)
    hip.check(lib.ams_k_depthwise3x3(P(xd), B, H, W, Cn, P(wd), stride, rate, PD(scale), PD(shift), hip.ACT_RELU6, P(y),
                                     stream()))
    xt = torch.as_tensor(x).double().permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.as_tensor(w).double().permute(2, 3, 0, 1).requires_grad_(True)
    raw = F.conv2d(F.pad(xt, (pl, pr, pt, pb)), wt, stride=stride, dilation=rate, groups=Cn)
    ref = torch.clamp(raw * torch.as_tensor(scale).view(1, -1, 1, 1) + torch.as_tensor(shift).view(1, -1, 1, 1), 0, 6)
    assert rel_err(y.cpu().numpy(), ref.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    # raw output (training-mode forward)
    hip.check(lib.ams_k_depthwise3x3(P(xd), B, H, W, Cn, P(wd), stride, rate, None, None, hip.ACT_NONE, P(y), stream()))
    assert rel_err(y.cpu().numpy(), raw.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    # backward
    dy = rng.standard_normal((B, Ho, Wo, Cn)).astype(np.float32)
    raw.backward(torch.as_tensor(dy).double().permute(0, 3, 1, 2))
    dx = torch.empty((B, H, W, Cn), device=DEV)
    hip.check(lib.ams_k_depthwise3x3_dgrad(PD(dy), B, H, W, Cn, P(wd), stride, rate, P(dx), stream()))
    assert rel_err(dx.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy()) < 1e-5
    dw = torch.empty((3, 3, Cn, 1), device=DEV)
    scr = torch.empty(1024 * 9 * Cn + 16, device=DEV)
    hip.check(lib.ams_k_depthwise3x3_wgrad(P(xd), PD(dy), B, H, W, Cn, stride, rate, P(dw), P(scr), scr.numel(), stream()))
    assert rel_err(dw.cpu().numpy(), wt.grad.permute(2, 3, 0, 1).numpy()) < 2e-5


@pytest.mark.parametrize("H,W,Cn,rate,act", [(33, 65, 384, 1, "relu6"), (33, 65, 960, 2, "relu6"), (17, 31, 576, 1, "relu6"), (9, 9, 64, 2, "none"),
                                             (20, 7, 36, 1, "relu6"), (257, 129, 32, 1, "relu6"), (5, 3, 1024, 2, "relu6")])
def test_depthwise_fine_tune_kernels(lib, H, W, Cn, rate, act):
    """The fine-tune step's one-kernel forward and backward of a stride-1 depthwise layer (k_conv.hip: dw3x3_fwd_bn_kernel,
    dw3x3_dgrad_bn_kernel) against f64 math on ragged sizes: result tensors, the BN sums folded from the partial rows, the nine weight-gradient
    taps.  Row bands (forward), column strips that wrap around the channel groups several times, more channels than a block has threads."""
    import ctypes as C
    rng = np.random.default_rng(H * W + Cn + rate)
    B = 2
    ze = rng.standard_normal((B, H, W, Cn)).astype(np.float32) * 2.0
    w = (rng.standard_normal((3, 3, Cn, 1)) * 0.4).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cn).astype(np.float32)
    shift = rng.standard_normal(Cn).astype(np.float32)
    center = rng.standard_normal(Cn).astype(np.float32) * 0.1
    mean = rng.standard_normal(Cn).astype(np.float32) * 0.2
    rstd = rng.uniform(0.5, 2.0, Cn).astype(np.float32)
    act_id = hip.ACT_RELU6 if act == "relu6" else hip.ACT_NONE
    _, pt, pb = S.same_pad(H, 3, 1, rate)
    _, pl, pr = S.same_pad(W, 3, 1, rate)
    zt = torch.as_tensor(ze).double().permute(0, 3, 1, 2)
    y = zt * torch.as_tensor(scale).double().view(1, -1, 1, 1) + torch.as_tensor(shift).double().view(1, -1, 1, 1)
    a_e = (torch.clamp(y, 0, 6) if act == "relu6" else y).requires_grad_(True)
    wt = torch.as_tensor(w).double().permute(2, 3, 0, 1).requires_grad_(True)
    zd_ref = F.conv2d(F.pad(a_e, (pl, pr, pt, pb)), wt, dilation=rate, groups=Cn)
    # ---- forward
    n_scr = lib.ams_k_depthwise3x3_fwd_bn_scratch(B, H, W, Cn, rate)
    scr = torch.full((n_scr,), float("nan"), device=DEV)
    zd = torch.empty((B, H, W, Cn), device=DEV)
    rows = C.c_int32(0)
    hip.check(lib.ams_k_depthwise3x3_fwd_bn(PD(ze), B, H, W, Cn, PD(w), rate, PD(scale), PD(shift), act_id, PD(center), P(zd), P(scr), n_scr,
                                            C.byref(rows), stream()))
    want = zd_ref.detach().permute(0, 2, 3, 1).numpy()
    assert rel_err(zd.cpu().numpy(), want) < 1e-5
    part = scr[: rows.value * 2 * Cn].cpu().numpy().astype(np.float64).reshape(rows.value, 2, Cn).sum(axis=0)
    d = want - center.astype(np.float64)
    assert rel_err(part[0], d.sum(axis=(0, 1, 2))) < 2e-5 and rel_err(part[1], (d * d).sum(axis=(0, 1, 2))) < 2e-5
    # ---- backward
    dz = rng.standard_normal((B, H, W, Cn)).astype(np.float32)
    zd_ref.backward(torch.as_tensor(dz).double().permute(0, 3, 1, 2))
    mask = ((y > 0) & (y < 6)).double() if act == "relu6" else torch.ones_like(y)
    dy_ref = (a_e.grad * mask).permute(0, 2, 3, 1).numpy()
    xhat = (ze.astype(np.float64) - mean) * rstd
    n_scr = lib.ams_k_depthwise3x3_dgrad_bn_scratch(B, H, W, Cn)
    scr = torch.full((n_scr,), float("nan"), device=DEV)
    out = torch.empty((B, H, W, Cn), device=DEV)
    hip.check(lib.ams_k_depthwise3x3_dgrad_bn(PD(dz), B, H, W, Cn, PD(w), rate, PD(ze), PD(scale), PD(shift), act_id, PD(mean), PD(rstd), P(out),
                                              P(scr), n_scr, C.byref(rows), stream()))
    assert rel_err(out.cpu().numpy(), dy_ref) < 1e-5
    part = scr[: rows.value * 11 * Cn].cpu().numpy().astype(np.float64).reshape(rows.value, 11, Cn).sum(axis=0)
    assert rel_err(part[0], dy_ref.sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(part[1], (dy_ref * xhat).sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(part[2:].reshape(3, 3, Cn), wt.grad.permute(2, 3, 0, 1).numpy()[..., 0]) < 2e-5


@pytest.mark.parametrize("H,W,Cn,rate,act", [(33, 65, 384, 1, "relu6"), (33, 65, 960, 2, "relu6"), (17, 31, 576, 1, "relu6"), (9, 9, 64, 2, "none"),
                                             (20, 7, 36, 1, "relu6"), (130, 70, 32, 1, "relu6"), (5, 3, 1024, 2, "relu6"), (4, 33, 100, 2, "relu6"),
                                             (34, 66, 960, 2, "relu6")])
def test_depthwise_backward_with_folded_apply(lib, H, W, Cn, rate, act):
    """k_dw_train.hip (AMS_OPT_FUSE_DGRAD_BN = 3): the depthwise layer's BN-backward apply pass dz_d = A dy + B + C z_d formed inside the
    one-kernel depthwise backward, through an LDS ring.  Against f64 math on ragged sizes (column strips of unequal width, a 1-column and a
    3-column map, odd and even sizes under rate 2 — the four parity classes have different extents —, a channel count that ends inside a
    64-channel chunk, row bands), and the written gradient BIT FOR BIT against the apply pass followed by dw3x3_dgrad_bn_kernel."""
    import ctypes as C
    rng = np.random.default_rng(H * W + Cn + rate + 17)
    B = 2
    ze = rng.standard_normal((B, H, W, Cn)).astype(np.float32) * 2.0
    w = (rng.standard_normal((3, 3, Cn, 1)) * 0.4).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cn).astype(np.float32)
    shift = rng.standard_normal(Cn).astype(np.float32)
    mean = rng.standard_normal(Cn).astype(np.float32) * 0.2
    rstd = rng.uniform(0.5, 2.0, Cn).astype(np.float32)
    dy_d = rng.standard_normal((B, H, W, Cn)).astype(np.float32)
    z_d = rng.standard_normal((B, H, W, Cn)).astype(np.float32)
    cA = rng.uniform(0.5, 1.5, Cn).astype(np.float32)
    cB = (rng.standard_normal(Cn) * 0.3).astype(np.float32)
    cC = (rng.standard_normal(Cn) * 0.3).astype(np.float32)
    act_id = hip.ACT_RELU6 if act == "relu6" else hip.ACT_NONE
    _, pt, pb = S.same_pad(H, 3, 1, rate)
    _, pl, pr = S.same_pad(W, 3, 1, rate)
    dz32 = ((cA * dy_d + cB) + cC * z_d).astype(np.float32)            # the apply pass's own f32 arithmetic
    dz64 = cA.astype(np.float64) * dy_d + cB + cC.astype(np.float64) * z_d
    zt = torch.as_tensor(ze).double().permute(0, 3, 1, 2)
    y = zt * torch.as_tensor(scale).double().view(1, -1, 1, 1) + torch.as_tensor(shift).double().view(1, -1, 1, 1)
    a_e = (torch.clamp(y, 0, 6) if act == "relu6" else y).requires_grad_(True)
    wt = torch.as_tensor(w).double().permute(2, 3, 0, 1).requires_grad_(True)
    zd_ref = F.conv2d(F.pad(a_e, (pl, pr, pt, pb)), wt, dilation=rate, groups=Cn)
    zd_ref.backward(torch.as_tensor(dz64).permute(0, 3, 1, 2))
    mask = ((y > 0) & (y < 6)).double() if act == "relu6" else torch.ones_like(y)
    dy_ref = (a_e.grad * mask).permute(0, 2, 3, 1).numpy()
    xhat = (ze.astype(np.float64) - mean) * rstd
    n_scr = lib.ams_k_depthwise3x3_dgrad_bn_apply_scratch(B, H, W, Cn, rate)
    scr = torch.full((n_scr,), float("nan"), device=DEV)
    out = torch.full((B, H, W, Cn), float("nan"), device=DEV)
    rows = C.c_int32(0)
    hip.check(lib.ams_k_depthwise3x3_dgrad_bn_apply(PD(dy_d), PD(z_d), PD(cA), PD(cB), PD(cC), B, H, W, Cn, PD(w), rate, PD(ze), PD(scale), PD(shift),
                                                    act_id, PD(mean), PD(rstd), P(out), P(scr), n_scr, C.byref(rows), stream()))
    got = out.cpu().numpy()
    assert np.isfinite(got).all(), "an output pixel was not written"
    assert rel_err(got, dy_ref) < 1e-5
    part = scr[: rows.value * 11 * Cn].cpu().numpy().astype(np.float64)
    assert np.isfinite(part).all(), "a partial-row entry was not written"
    part = part.reshape(rows.value, 11, Cn).sum(axis=0)
    assert rel_err(part[0], dy_ref.sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(part[1], (dy_ref * xhat).sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(part[2:].reshape(3, 3, Cn), wt.grad.permute(2, 3, 0, 1).numpy()[..., 0]) < 2e-5
    # ---- the forward in the same tile form: against f64 math, and z_d bit for bit against dw3x3_fwd_bn_kernel
    center = (rng.standard_normal(Cn) * 0.1).astype(np.float32)
    n_f = lib.ams_k_depthwise3x3_fwd_bn_tiles_scratch(B, H, W, Cn, rate)
    scr_f = torch.full((n_f,), float("nan"), device=DEV)
    zd_t = torch.full((B, H, W, Cn), float("nan"), device=DEV)
    hip.check(lib.ams_k_depthwise3x3_fwd_bn_tiles(PD(ze), B, H, W, Cn, PD(w), rate, PD(scale), PD(shift), act_id, PD(center), P(zd_t), P(scr_f), n_f,
                                                  C.byref(rows), stream()))
    want_zd = zd_ref.detach().permute(0, 2, 3, 1).numpy()
    assert np.isfinite(zd_t.cpu().numpy()).all() and rel_err(zd_t.cpu().numpy(), want_zd) < 1e-5
    part_f = scr_f[: rows.value * 2 * Cn].cpu().numpy().astype(np.float64)
    assert np.isfinite(part_f).all()
    part_f = part_f.reshape(rows.value, 2, Cn).sum(axis=0)
    dctr = want_zd - center.astype(np.float64)
    assert rel_err(part_f[0], dctr.sum(axis=(0, 1, 2))) < 2e-5 and rel_err(part_f[1], (dctr * dctr).sum(axis=(0, 1, 2))) < 2e-5
    if Cn <= 1024:
        n_o = lib.ams_k_depthwise3x3_fwd_bn_scratch(B, H, W, Cn, rate)
        scr_o = torch.empty(n_o, device=DEV)
        zd_o = torch.empty((B, H, W, Cn), device=DEV)
        hip.check(lib.ams_k_depthwise3x3_fwd_bn(PD(ze), B, H, W, Cn, PD(w), rate, PD(scale), PD(shift), act_id, PD(center), P(zd_o), P(scr_o), n_o,
                                                C.byref(rows), stream()))
        assert torch.equal(zd_t, zd_o), "tile-form forward differs from dw3x3_fwd_bn_kernel by %g" % (zd_t - zd_o).abs().max().item()
    # the pass it replaces + the kernel it replaces: same bits in the written gradient
    if Cn <= 1024:
        n2 = lib.ams_k_depthwise3x3_dgrad_bn_scratch(B, H, W, Cn)
        scr2 = torch.empty(n2, device=DEV)
        out2 = torch.empty((B, H, W, Cn), device=DEV)
        hip.check(lib.ams_k_depthwise3x3_dgrad_bn(PD(dz32), B, H, W, Cn, PD(w), rate, PD(ze), PD(scale), PD(shift), act_id, PD(mean), PD(rstd), P(out2),
                                                  P(scr2), n2, C.byref(rows), stream()))
        assert torch.equal(out, out2), "folded apply differs from the separate pass by %g" % (out - out2).abs().max().item()


@pytest.mark.parametrize("H,W,Cin,Cexp,stride", [(20, 37, 24, 144, 1), (21, 38, 24, 144, 2), (22, 37, 24, 144, 2), (16, 16, 16, 96, 2),
                                                 (9, 50, 32, 192, 1), (33, 65, 32, 192, 2), (8, 8, 8, 32, 1), (5, 3, 12, 48, 2), (64, 128, 16, 96, 1)])
def test_recompute_block_kernels(lib, H, W, Cin, Cexp, stride):
    """The fine-tune step of an early block without its expanded tensors (k_xdw_train.hip) through its own C-ABI entries, against f64
    autograd of the materialised block: forward statistics + x^T x + sum x, the backward partial rows (BN sums, depthwise taps, x^T dy),
    the input gradient with the skip gradient, and the expand weight gradient rebuilt from the reduced sums.  Odd and even sizes at stride 2
    (SAME pads 0 or 1 before: the parity classes of the transposed conv move), maps smaller than a tile, Cin below a full k chunk."""
    import ctypes as C
    rng = np.random.default_rng(H * 131 + W * 7 + Cin + stride)
    B = 2
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    we = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, Cexp, 1)) * 0.4).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    sh = (rng.standard_normal(Cexp) + 1.0).astype(np.float32)
    mean = (rng.standard_normal(Cexp) * 0.2).astype(np.float32)
    rstd = rng.uniform(0.5, 2.0, Cexp).astype(np.float32)
    center = (rng.standard_normal(Cexp) * 0.1).astype(np.float32)
    cA, cB, cC = (rng.standard_normal(Cexp).astype(np.float32) for _ in range(3))
    Ho, pt, pb = S.same_pad(H, 3, stride, 1)
    Wo, pl, pr = S.same_pad(W, 3, stride, 1)
    dz_d = rng.standard_normal((B, Ho, Wo, Cexp)).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    KP = (Cin + 15) // 16 * 16
    # ---- f64 reference of the materialised block
    xt = torch.as_tensor(x).double()
    wet = torch.as_tensor(we).double()
    ze = xt @ wet                                                        # [B,H,W,Cexp]
    y = ze * torch.as_tensor(sc).double() + torch.as_tensor(sh).double()
    ae = torch.clamp(y, 0, 6).permute(0, 3, 1, 2).requires_grad_(True)
    wdt = torch.as_tensor(wd).double().permute(2, 3, 0, 1).requires_grad_(True)
    zd = F.conv2d(F.pad(ae, (pl, pr, pt, pb)), wdt, stride=stride, groups=Cexp)
    zd.backward(torch.as_tensor(dz_d).double().permute(0, 3, 1, 2))
    dy = (ae.grad.permute(0, 2, 3, 1) * ((y > 0) & (y < 6)).double()).numpy()       # [B,H,W,Cexp]
    zen = ze.numpy()
    xhat = (zen - mean.astype(np.float64)) * rstd.astype(np.float64)
    x64 = x.astype(np.float64).reshape(-1, Cin)
    n_scr = lib.ams_k_xdw_train_scratch(B, H, W, Cin, Cexp)
    rows, stride_out = C.c_int32(0), C.c_int64(0)
    # ---- forward statistics
    scr = torch.full((n_scr,), float("nan"), device=DEV)
    hip.check(lib.ams_k_xdw_fwd_stats(PD(x), B, H, W, Cin, PD(we), Cexp, PD(center), P(scr), n_scr, C.byref(rows), C.byref(stride_out), stream()))
    part = scr[: rows.value * stride_out.value].cpu().numpy().astype(np.float64).reshape(rows.value, stride_out.value).sum(axis=0)
    d = zen - center.astype(np.float64)
    assert rel_err(part[:Cexp], d.sum(axis=(0, 1, 2))) < 2e-5 and rel_err(part[Cexp:2 * Cexp], (d * d).sum(axis=(0, 1, 2))) < 2e-5
    XX = part[2 * Cexp:2 * Cexp + KP * KP].reshape(KP, KP)
    g0 = part[2 * Cexp + KP * KP:2 * Cexp + KP * KP + KP]
    assert rel_err(XX[:Cin, :Cin], x64.T @ x64) < 2e-5 and rel_err(g0[:Cin], x64.sum(axis=0)) < 2e-5
    # ---- backward: partial rows
    scr2 = torch.full((n_scr,), float("nan"), device=DEV)
    hip.check(lib.ams_k_xdw_bwd_reduce(PD(x), B, H, W, Cin, PD(we), Cexp, PD(sc), PD(sh), PD(mean), PD(rstd), hip.ACT_RELU6, PD(wd), stride, PD(dz_d),
                                       P(scr2), n_scr, C.byref(rows), C.byref(stride_out), stream()))
    pb_ = scr2[: rows.value * stride_out.value].cpu().numpy().astype(np.float64).reshape(rows.value, stride_out.value).sum(axis=0)
    assert rel_err(pb_[:Cexp], dy.sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(pb_[Cexp:2 * Cexp], (dy * xhat).sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(pb_[2 * Cexp:11 * Cexp].reshape(3, 3, Cexp), wdt.grad.permute(2, 3, 0, 1).numpy()[..., 0]) < 2e-5
    G1 = pb_[11 * Cexp:11 * Cexp + KP * Cexp].reshape(KP, Cexp)
    assert rel_err(G1[:Cin], x64.T @ dy.reshape(-1, Cexp)) < 2e-5
    # ---- backward: input gradient
    dze = cA.astype(np.float64) * dy + cB.astype(np.float64) + cC.astype(np.float64) * zen
    dx_ref = dze.reshape(-1, Cexp) @ we.astype(np.float64).T + res.astype(np.float64).reshape(-1, Cin)
    dx = torch.empty((B, H, W, Cin), device=DEV)
    hip.check(lib.ams_k_xdw_bwd_dx(PD(x), B, H, W, Cin, PD(we), Cexp, PD(sc), PD(sh), hip.ACT_RELU6, PD(wd), stride, PD(dz_d), PD(cA), PD(cB), PD(cC),
                                   PD(res), P(dx), stream()))
    assert rel_err(dx.cpu().numpy().reshape(-1, Cin), dx_ref) < 2e-5
    # ---- expand weight gradient from the reduced sums: x^T dz_e without a pass over dz_e
    xx_g0 = np.concatenate([XX.reshape(-1), g0]).astype(np.float32)
    dwe = torch.empty((Cin, Cexp), device=DEV)
    hip.check(lib.ams_k_xdw_dwe(PD(G1.astype(np.float32)), PD(xx_g0), Cin, Cexp, PD(we), PD(cA), PD(cB), PD(cC), P(dwe), stream()))
    assert rel_err(dwe.cpu().numpy(), x64.T @ dze.reshape(-1, Cexp)) < 5e-5


@pytest.mark.parametrize("H,W,dtype", [(32, 64, np.uint8), (37, 50, np.uint8), (9, 7, np.float32), (128, 256, np.uint8)])
def test_recompute_first_block_kernel(lib, H, W, dtype):
    """The first block's backward without the stem's tensors (launch_xdw_bwd_reduce_stem): BN-backward sums of the stem, the first depthwise
    layer's weight gradient and the pieces of the stem weight gradient (x^T dy over the 27-tap patch, x^T x, sum x) against f64 autograd."""
    import ctypes as C
    rng = np.random.default_rng(H + 3 * W)
    B = 2
    frames = rng.integers(0, 256, (B, H, W, 3)).astype(dtype)
    ws = (rng.standard_normal((3, 3, 3, 32)) * 0.3).astype(np.float32)
    wd = (rng.standard_normal((3, 3, 32, 1)) * 0.4).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    sh = (rng.standard_normal(32) + 1.0).astype(np.float32)
    mean = (rng.standard_normal(32) * 0.2).astype(np.float32)
    rstd = rng.uniform(0.5, 2.0, 32).astype(np.float32)
    H1, pt, pb = S.same_pad(H + 1, 3, 2, 1)
    W1, pl, pr = S.same_pad(W + 1, 3, 2, 1)
    dz_d = rng.standard_normal((B, H1, W1, 32)).astype(np.float32)
    # reference: normalised, 127.5-padded frame -> im2col patches [B,H1,W1,27] (k = tap * 3 + channel) -> 1x1 conv -> BN -> ReLU6 -> depthwise
    xin = torch.as_tensor(frames.astype(np.float32)).permute(0, 3, 1, 2)
    xin = (F.pad(xin, (0, 1, 0, 1), value=127.5) * np.float32(S.PIXEL_SCALE) - 1.0).double()
    xp = F.pad(xin, (pl, pr, pt, pb))
    patches = F.unfold(xp, kernel_size=3, stride=2).view(B, 3, 9, H1, W1).permute(0, 3, 4, 2, 1).reshape(B, H1, W1, 27)     # [.., tap, ch]
    wmat = torch.as_tensor(ws).double().reshape(27, 32)                                                               # HWIO: (tap, ch) rows
    ze = patches @ wmat
    y = ze * torch.as_tensor(sc).double() + torch.as_tensor(sh).double()
    ae = torch.clamp(y, 0, 6).permute(0, 3, 1, 2).requires_grad_(True)
    wdt = torch.as_tensor(wd).double().permute(2, 3, 0, 1).requires_grad_(True)
    zd = F.conv2d(F.pad(ae, (1, 1, 1, 1)), wdt, groups=32)
    zd.backward(torch.as_tensor(dz_d).double().permute(0, 3, 1, 2))
    dy = (ae.grad.permute(0, 2, 3, 1) * ((y > 0) & (y < 6)).double()).numpy()
    xhat = (ze.numpy() - mean.astype(np.float64)) * rstd.astype(np.float64)
    p64 = patches.numpy().reshape(-1, 27)
    n_scr = lib.ams_k_xdw_stem_scratch(B, H, W)
    scr = torch.full((n_scr,), float("nan"), device=DEV)
    rows, stride_out = C.c_int32(0), C.c_int64(0)
    fd = torch.as_tensor(frames).to(DEV)
    hip.check(lib.ams_k_xdw_bwd_reduce_stem(P(fd), hip.DT_U8 if dtype == np.uint8 else hip.DT_F32, B, H, W, S.PIXEL_SCALE, PD(ws), PD(sc), PD(sh),
                                            PD(mean), PD(rstd), hip.ACT_RELU6, PD(wd), PD(dz_d), P(scr), n_scr, C.byref(rows), C.byref(stride_out),
                                            stream()))
    part = scr[: rows.value * stride_out.value].cpu().numpy().astype(np.float64).reshape(rows.value, stride_out.value).sum(axis=0)
    assert rel_err(part[:32], dy.sum(axis=(0, 1, 2))) < 2e-5 and rel_err(part[32:64], (dy * xhat).sum(axis=(0, 1, 2))) < 2e-5
    assert rel_err(part[64:11 * 32].reshape(3, 3, 32), wdt.grad.permute(2, 3, 0, 1).numpy()[..., 0]) < 2e-5
    G1 = part[11 * 32:11 * 32 + 32 * 32].reshape(32, 32)
    assert rel_err(G1[:27], p64.T @ dy.reshape(-1, 32)) < 2e-5
    base = 11 * 32 + 32 * 32
    XX = part[base:base + 32 * 32].reshape(32, 32)
    g0 = part[base + 32 * 32:base + 32 * 32 + 32]
    assert rel_err(XX[:27, :27], p64.T @ p64) < 2e-5 and rel_err(g0[:27], p64.sum(axis=0)) < 2e-5


@pytest.mark.parametrize("H,W,Cin,Cexp,stride", [(33, 65, 16, 96, 2), (40, 37, 24, 144, 1), (33, 65, 24, 144, 2), (29, 50, 32, 192, 1),
                                                 (17, 17, 64, 384, 1), (65, 129, 16, 96, 1)])
def test_fused_expand_depthwise(lib, H, W, Cin, Cexp, stride):
    rng = np.random.default_rng(H + Cin)
    B = 2
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    we = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, Cexp, 1)) * 0.4).astype(np.float32)
    se, sd = rng.uniform(0.5, 1.5, Cexp).astype(np.float32), rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    he, hd = rng.standard_normal(Cexp).astype(np.float32), rng.standard_normal(Cexp).astype(np.float32)
    Ho, pt, pb = S.same_pad(H, 3, stride, 1)
    Wo, pl, pr = S.same_pad(W, 3, stride, 1)
    y = torch.full((B, Ho, Wo, Cexp), np.nan, device=DEV)
    hip.check(lib.ams_k_expand_dw(PD(x), B, H, W, Cin, PD(we), PD(se), PD(he), Cexp, PD(wd), stride, 1, PD(sd), PD(hd), P(y), stream()))
    e = np.clip((x.astype(np.float64) @ we.astype(np.float64)) * se + he, 0, 6)
    et = torch.as_tensor(e).permute(0, 3, 1, 2)
    raw = F.conv2d(F.pad(et, (pl, pr, pt, pb)), torch.as_tensor(wd).double().permute(2, 3, 0, 1), stride=stride, groups=Cexp)
    ref = torch.clamp(raw * torch.as_tensor(sd).view(1, -1, 1, 1) + torch.as_tensor(hd).view(1, -1, 1, 1), 0, 6)
    assert rel_err(y.cpu().numpy(), ref.permute(0, 2, 3, 1).numpy()) < 2e-5


@pytest.mark.parametrize("H,W,Cin,Cexp,Cout,stride,res", [(33, 65, 16, 96, 24, 2, False), (40, 37, 24, 144, 24, 1, True), (33, 65, 24, 144, 32, 2, False),
                                                          (29, 50, 32, 192, 32, 1, True), (34, 66, 32, 192, 64, 2, False), (65, 129, 16, 96, 24, 1, False),
                                                          (7, 5, 24, 144, 24, 1, True), (16, 16, 32, 192, 32, 1, False)])
def test_whole_block_kernel(lib, H, W, Cin, Cexp, Cout, stride, res, knobs):
    """k_block.hip: expand + depthwise + project (+ block input) in one kernel.  Against f64 math, and bit for bit against the
    kernels it replaces (fused expand+depthwise, then the f32 GEMM with its epilogue) — same products, same k order."""
    rng = np.random.default_rng(H * 3 + Cin + Cout)
    B = 2
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    we = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, Cexp, 1)) * 0.4).astype(np.float32)
    wp = (rng.standard_normal((Cexp, Cout)) / np.sqrt(Cexp)).astype(np.float32)
    se, sd, sp = (rng.uniform(0.5, 1.5, n).astype(np.float32) for n in (Cexp, Cexp, Cout))
    he, hd, hp = (rng.standard_normal(n).astype(np.float32) for n in (Cexp, Cexp, Cout))
    Ho, pt, pb = S.same_pad(H, 3, stride, 1)
    Wo, pl, pr = S.same_pad(W, 3, stride, 1)
    xd, wed, wdd, wpd = dev(x), dev(we), dev(wd), dev(wp)
    sed, hed, sdd, hdd, spd, hpd = dev(se), dev(he), dev(sd), dev(hd), dev(sp), dev(hp)
    y = torch.full((B, Ho, Wo, Cout), np.nan, device=DEV)
    hip.check(lib.ams_k_block_fused(P(xd), B, H, W, Cin, P(wed), P(sed), P(hed), Cexp, P(wdd), stride, P(sdd), P(hdd), P(wpd), Cout, P(spd), P(hpd),
                                    int(res), P(y), None, 0, stream()))
    e = np.clip((x.astype(np.float64) @ we.astype(np.float64)) * se + he, 0, 6)
    et = torch.as_tensor(e).permute(0, 3, 1, 2)
    raw = F.conv2d(F.pad(et, (pl, pr, pt, pb)), torch.as_tensor(wd).double().permute(2, 3, 0, 1), stride=stride, groups=Cexp)
    d = torch.clamp(raw * torch.as_tensor(sd).view(1, -1, 1, 1) + torch.as_tensor(hd).view(1, -1, 1, 1), 0, 6).permute(0, 2, 3, 1).numpy()
    ref = (d @ wp.astype(np.float64)) * sp + hp
    if res:
        ref = ref + x
    got = y.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, ref) < 2e-5
    # the two kernels it replaces
    dmid = torch.empty((B, Ho, Wo, Cexp), device=DEV)
    hip.check(lib.ams_k_expand_dw(P(xd), B, H, W, Cin, P(wed), P(sed), P(hed), Cexp, P(wdd), stride, 1, P(sdd), P(hdd), P(dmid), stream()))
    y2 = torch.empty((B, Ho, Wo, Cout), device=DEV)
    hip.check(lib.ams_k_pointwise(P(dmid), B * Ho * Wo, Cexp, P(wpd), Cout, 0, None, 1, P(spd), P(hpd), hip.ACT_NONE, P(xd) if res else None, P(y2),
                                  stream()))
    assert torch.equal(y, y2), "max abs diff %g" % (y - y2).abs().max().item()
    # the default form of the engine for K >= 24: expand products as six bf16 MFMAs on three-part splits (f32-level, not the same bits)
    panels = torch.zeros(3 * Cexp * 32, dtype=torch.int16, device=DEV)
    y3 = torch.full((B, Ho, Wo, Cout), np.nan, device=DEV)
    hip.check(lib.ams_k_block_fused(P(xd), B, H, W, Cin, P(wed), P(sed), P(hed), Cexp, P(wdd), stride, P(sdd), P(hdd), P(wpd), Cout, P(spd), P(hpd),
                                    int(res), P(y3), P(panels), panels.numel(), stream()))
    assert rel_err(y3.cpu().numpy(), ref) < 2e-5
    if Cin > 16:
        assert not torch.equal(y3, y) and rel_err(y3.cpu().numpy(), got) < 2e-6
    else:
        assert torch.equal(y3, y)                      # K = 16: the exact-f32 form either way
    # AMS_MATMUL_SPLIT_F16: expand and project products on two fp16 parts (3 MFMAs each, K = 16 included) — f32-level; tile knobs change no bit
    hpan = torch.zeros(2 * Cexp * 32 + 2 * Cout * ((Cexp + 31) // 32 * 32), dtype=torch.int16, device=DEV)
    outs = []
    for tile in (None, "4x8", "2x8" if stride == 2 else "8x8"):
        knobs(AMS_BLK_TILE=tile)
        y4 = torch.full((B, Ho, Wo, Cout), np.nan, device=DEV)
        hip.check(lib.ams_k_block_fused_f16(P(xd), B, H, W, Cin, P(wed), P(sed), P(hed), Cexp, P(wdd), stride, P(sdd), P(hdd), P(wpd), Cout, P(spd), P(hpd),
                                            int(res), P(y4), P(hpan), hpan.numel(), stream()))
        assert rel_err(y4.cpu().numpy(), ref) < 2e-5 and rel_err(y4.cpu().numpy(), got) < 4e-6, tile
        outs.append(y4)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("H,W,Cin,Cexp,rate,parts", [(33, 65, 64, 384, 1, 3), (33, 65, 96, 576, 1, 2), (33, 65, 160, 960, 2, 3),
                                                     (17, 33, 160, 960, 2, 2), (9, 200, 64, 384, 1, 3), (5, 3, 96, 576, 2, 3),
                                                     (40, 7, 64, 96, 1, 2), (2, 2, 160, 320, 2, 3), (1, 70, 96, 192, 1, 3),
                                                     (33, 129, 32, 192, 1, 3), (12, 19, 32, 64, 1, 2), (37, 70, 24, 144, 1, 3),
                                                     (20, 33, 16, 96, 1, 3)])
def test_fused_expand_depthwise_stream(lib, H, W, Cin, Cexp, rate, parts, knobs):
    """Streaming expand+depthwise vs f64, and bit-for-bit against the two kernels it replaces (Cin <= 32: exact-f32 products,
    compared with the f32 GEMM; Cin >= 64: split-bf16 products, compared with the split GEMM of the same number of parts).  Ragged sizes: column strips (W = 200), images
    smaller than one step, single rows, sub-images of unequal size (odd H, W at rate 2); every tile / segment geometry."""
    rng = np.random.default_rng(H * 7 + W + Cin + rate)
    B = 3
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    we = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, Cexp, 1)) * 0.4).astype(np.float32)
    se, sd = rng.uniform(0.5, 1.5, Cexp).astype(np.float32), rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    he, hd = rng.standard_normal(Cexp).astype(np.float32), rng.standard_normal(Cexp).astype(np.float32)
    panels = torch.zeros(3 * Cexp * Cin + 3 * B * H * W * Cin, dtype=torch.int16, device=DEV)
    e = np.clip((x.astype(np.float64) @ we.astype(np.float64)) * se + he, 0, 6)
    et = torch.as_tensor(e).permute(0, 3, 1, 2)
    raw = F.conv2d(F.pad(et, (rate, rate, rate, rate)), torch.as_tensor(wd).double().permute(2, 3, 0, 1), dilation=rate, groups=Cexp)
    ref = torch.clamp(raw * torch.as_tensor(sd).view(1, -1, 1, 1) + torch.as_tensor(hd).view(1, -1, 1, 1), 0, 6).permute(0, 2, 3, 1).numpy()
    unfused = None
    if True:
        M = B * H * W
        ebuf = torch.empty((M, Cexp), device=DEV)
        if Cin <= 32:
            hip.check(lib.ams_k_pointwise(PD(x), M, Cin, PD(we), Cexp, 0, None, 1, PD(se), PD(he), hip.ACT_RELU6, None, P(ebuf), stream()))
        else:
            hip.check((lib.ams_k_pointwise_split if parts == 2 else lib.ams_k_pointwise_split3)(PD(x), M, Cin, PD(we), Cexp, PD(se), PD(he), hip.ACT_RELU6, None,
                                                                                               P(ebuf), P(panels), panels.numel(), stream()))
        unfused = torch.empty((B, H, W, Cexp), device=DEV)
        hip.check(lib.ams_k_depthwise3x3(P(ebuf), B, H, W, Cexp, PD(wd), 1, rate, PD(sd), PD(hd), hip.ACT_RELU6, P(unfused), stream()))
        unfused = unfused.cpu().numpy()
    # (16-channel tiles per block, row segments, column strips, E-waves, D-waves, blocks per channel chunk)
    for force in (None, "4,1,1", "2,3,2", "4,2,3", "2,1,1,4,4,5", "2,2,1,8,4,3", "4,1,2,8,4,1"):
        # operand split inside the kernel / loaded as bf16 parts (what the previous block's GEMM writes) / the weight-register
        # form of the kernel (k_xdw_wreg.hip: AMS_XWR_FORCE = E-waves, row segments, column strips, blocks per channel group,
        # row groups per E-wave and step)
        knobs(AMS_XDS_FORCE=force, AMS_XWR_FORCE={None: "4,0,0,0,2", "4,1,1": "8,1,1,0,1", "2,3,2": "4,3,2,0,1", "4,2,3": "8,2,3,2,1"}.get(force, "4,2,1,3,2"))
        for pre in ((0, 1, 2) if Cin >= 64 else (0,)):      # pre-split operands and the weight-register form: split products only
            y = torch.full((B, H, W, Cexp), np.nan, device=DEV)
            hip.check(lib.ams_k_expand_dw_stream(PD(x), B, H, W, Cin, PD(we), PD(se), PD(he), Cexp, PD(wd), rate, PD(sd), PD(hd), P(y),
                                                 P(panels), panels.numel(), parts, pre, stream()))
            got = y.cpu().numpy()
            assert np.isfinite(got).all(), (force, pre)
            assert rel_err(got, ref) < (5e-5 if parts == 2 and Cin >= 64 else 2e-5), (force, pre)
            assert np.array_equal(got, unfused), (force, pre)


@pytest.mark.parametrize("H,W,Cin,Cexp", [(65, 129, 16, 96), (64, 130, 24, 144), (33, 40, 32, 192), (7, 9, 16, 32), (2, 3, 24, 48)])
def test_fused_expand_depthwise_stream_stride2(lib, H, W, Cin, Cexp, knobs):
    """Stride-2 blocks on the streaming kernel (exact-f32 products): vs f64 and bit-for-bit against f32 GEMM + stride-2 depthwise.
    Odd and even sizes (SAME padding puts the window centres on even or odd input positions)."""
    rng = np.random.default_rng(H * 5 + W + Cin)
    B = 3
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    we = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, Cexp, 1)) * 0.4).astype(np.float32)
    se, sd = rng.uniform(0.5, 1.5, Cexp).astype(np.float32), rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    he, hd = rng.standard_normal(Cexp).astype(np.float32), rng.standard_normal(Cexp).astype(np.float32)
    Ho, pt, pb = S.same_pad(H, 3, 2, 1)
    Wo, pl, pr = S.same_pad(W, 3, 2, 1)
    e = np.clip((x.astype(np.float64) @ we.astype(np.float64)) * se + he, 0, 6)
    et = torch.as_tensor(e).permute(0, 3, 1, 2)
    raw = F.conv2d(F.pad(et, (pl, pr, pt, pb)), torch.as_tensor(wd).double().permute(2, 3, 0, 1), stride=2, groups=Cexp)
    ref = torch.clamp(raw * torch.as_tensor(sd).view(1, -1, 1, 1) + torch.as_tensor(hd).view(1, -1, 1, 1), 0, 6).permute(0, 2, 3, 1).numpy()
    M = B * H * W
    ebuf = torch.empty((M, Cexp), device=DEV)
    hip.check(lib.ams_k_pointwise(PD(x), M, Cin, PD(we), Cexp, 0, None, 1, PD(se), PD(he), hip.ACT_RELU6, None, P(ebuf), stream()))
    unfused = torch.empty((B, Ho, Wo, Cexp), device=DEV)
    hip.check(lib.ams_k_depthwise3x3(P(ebuf), B, H, W, Cexp, PD(wd), 2, 1, PD(sd), PD(hd), hip.ACT_RELU6, P(unfused), stream()))
    unfused = unfused.cpu().numpy()
    for force in (None, "4,1,1", "2,3,2", "4,2,3", "2,5,1"):
        knobs(AMS_XDS_FORCE=force)
        y = torch.full((B, Ho, Wo, Cexp), np.nan, device=DEV)
        hip.check(lib.ams_k_expand_dw_stream(PD(x), B, H, W, Cin, PD(we), PD(se), PD(he), Cexp, PD(wd), -2, PD(sd), PD(hd), P(y), None, 0, 3, 0,
                                             stream()))
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), force
        assert rel_err(got, ref) < 2e-5, force
        assert np.array_equal(got, unfused), force


def f16_parts(a):
    """the two fp16 parts of split_bf16.hpp on the host: hi = f16(x), lo = f16((x - hi) 2^11), both round-to-nearest-even"""
    a = np.asarray(a, dtype=np.float32)
    hi = a.astype(np.float16)
    lo = ((a - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def h2i_pack(a):
    """[M, C] f32 -> the H2I bytes as a uint16 array [M, C / 8, 2, 8] (per 8 channels: 8 hi, then 8 lo)"""
    hi, lo = f16_parts(a)
    M, C_ = a.shape
    return np.stack([hi.reshape(M, C_ // 8, 8), lo.reshape(M, C_ // 8, 8)], axis=2).view(np.uint16)


@pytest.mark.parametrize("M,K,N,res", [(2145, 960, 160, True), (17160, 160, 960, False), (2145, 384, 64, True), (4290, 576, 160, False),
                                       (2145, 256, 19, False), (1000, 64, 384, False), (300, 320, 256, False), (68640 // 4 + 7, 960, 320, False),
                                       (33, 96, 576, False), (5000, 576, 96, True), (129, 40, 24, False)])
def test_pointwise_split_f16(lib, M, K, N, res, knobs):
    """Two-fp16-part product (3 MFMAs per 32 k): f32-level against f64 (per product ~3 2^-22: 2e-6 of the output scale, the three-part bf16
    form's bar); the operand pre-packed as fp16 pairs gives the same bits as the in-kernel split; the optional part planes of the result
    equal the host's split of the f32 result; activations in [0, 6] like the depthwise results and N(0, 1) ones like block inputs; every
    tile shape the launcher can pick and half-height tail blocks."""
    rng = np.random.default_rng(M + K + N)
    x = (np.clip(rng.standard_normal((M, K)) * 2 + 1, 0, 6) if res else rng.standard_normal((M, K))).astype(np.float32)
    x[0, :8] = [0.0, 6.0, 1e-7, 3e-5, 6.1e-5, 1.0, 0.333333, 5.9999995]          # fp16 subnormal / boundary values of hi and lo
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32)
    shift = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32) if res else None
    Kp = (K + 31) // 32 * 32
    panels = torch.zeros(2 * N * Kp, dtype=torch.int16, device=DEV)
    want = (x.astype(np.float64) @ w.astype(np.float64)) * scale + shift + (r if res else 0.0)
    vec = N % 4 == 0
    outs = []
    for force in (None, "1,2", "2,3", "1,5", "2,4", "2,6"):
        knobs(AMS_PWX_FORCE=force)
        y = torch.full((M, N), np.nan, device=DEV)
        parts = torch.zeros((2, M, N), dtype=torch.int16, device=DEV) if vec else None
        hip.check(lib.ams_k_pointwise_split_f16(PD(x), M, K, PD(w), N, PD(scale), PD(shift), hip.ACT_NONE, PD(r) if res else None, P(y), P(panels),
                                                panels.numel(), None, P(parts), stream()))
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), force
        assert rel_err(got, want) < 2e-6, force
        outs.append(got)
        if vec:
            hi, lo = f16_parts(got)
            pp = parts.cpu().numpy().view(np.uint16)
            assert np.array_equal(pp[0], hi.view(np.uint16)) and np.array_equal(pp[1], lo.view(np.uint16)), force
        if K % 8 == 0:
            xh = torch.full((M, K), np.nan, device=DEV)
            y2 = torch.full((M, N), np.nan, device=DEV)
            hip.check(lib.ams_k_pointwise_split_f16(PD(x), M, K, PD(w), N, PD(scale), PD(shift), hip.ACT_NONE, PD(r) if res else None, P(y2), P(panels),
                                                    panels.numel(), P(xh), None, stream()))
            assert np.array_equal(xh.cpu().numpy().view(np.uint16).reshape(M, K // 8, 2, 8), h2i_pack(x)), force
            assert torch.equal(y2, y), force
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])              # same products in the same order whatever the tile


@pytest.mark.parametrize("H,W,Cin,Cexp,rate", [(33, 65, 64, 384, 1), (33, 65, 96, 576, 1), (33, 65, 160, 960, 2), (17, 33, 160, 960, 2),
                                               (9, 200, 64, 384, 1), (5, 3, 96, 576, 2), (40, 7, 64, 96, 1), (2, 2, 160, 320, 2), (1, 70, 96, 192, 1)])
def test_fused_expand_depthwise_stream_f16(lib, H, W, Cin, Cexp, rate, knobs):
    """The streaming expand + depthwise kernels on two fp16 parts: against f64, bit for bit against ams_k_pointwise_split_f16 followed by the
    depthwise kernel, and — written as fp16 pairs (y_h2i) — bit for bit the host's packing of that result.  Geometries as in the bf16 test."""
    rng = np.random.default_rng(H * 7 + W + Cin + rate)
    B = 3
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    we = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, Cexp, 1)) * 0.4).astype(np.float32)
    se, sd = rng.uniform(0.5, 1.5, Cexp).astype(np.float32), rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    he, hd = rng.standard_normal(Cexp).astype(np.float32), rng.standard_normal(Cexp).astype(np.float32)
    panels = torch.zeros(2 * Cexp * Cin + 2 * B * H * W * Cin, dtype=torch.int16, device=DEV)
    e = np.clip((x.astype(np.float64) @ we.astype(np.float64)) * se + he, 0, 6)
    et = torch.as_tensor(e).permute(0, 3, 1, 2)
    raw = F.conv2d(F.pad(et, (rate, rate, rate, rate)), torch.as_tensor(wd).double().permute(2, 3, 0, 1), dilation=rate, groups=Cexp)
    ref = torch.clamp(raw * torch.as_tensor(sd).view(1, -1, 1, 1) + torch.as_tensor(hd).view(1, -1, 1, 1), 0, 6).permute(0, 2, 3, 1).numpy()
    M = B * H * W
    ebuf = torch.empty((M, Cexp), device=DEV)
    hip.check(lib.ams_k_pointwise_split_f16(PD(x), M, Cin, PD(we), Cexp, PD(se), PD(he), hip.ACT_RELU6, None, P(ebuf), P(panels), panels.numel(), None,
                                            None, stream()))
    unfused = torch.empty((B, H, W, Cexp), device=DEV)
    hip.check(lib.ams_k_depthwise3x3(P(ebuf), B, H, W, Cexp, PD(wd), 1, rate, PD(sd), PD(hd), hip.ACT_RELU6, P(unfused), stream()))
    unfused = unfused.cpu().numpy()
    packed = h2i_pack(unfused.reshape(M, Cexp))
    for force in (None, "4,1,1", "2,3,2", "4,2,3", "2,1,1,4,4,5"):
        knobs(AMS_XDS_FORCE=force, AMS_XWR_FORCE={None: "4,0,0,0,2", "4,1,1": "8,1,1,0,1", "2,3,2": "4,3,2,0,1", "4,2,3": "8,2,3,2,1"}.get(force, "4,2,1,3,2"))
        for pre in (0, 1, 2):
            for y_h2i in (0, 1):
                y = torch.full((B, H, W, Cexp), np.nan, device=DEV)
                hip.check(lib.ams_k_expand_dw_stream_f16(PD(x), B, H, W, Cin, PD(we), PD(se), PD(he), Cexp, PD(wd), rate, PD(sd), PD(hd), P(y),
                                                         P(panels), panels.numel(), pre, y_h2i, stream()))
                got = y.cpu().numpy()
                if y_h2i:
                    assert np.array_equal(got.view(np.uint16).reshape(M, Cexp // 8, 2, 8), packed), (force, pre)
                else:
                    assert np.isfinite(got).all(), (force, pre)
                    assert rel_err(got, ref) < 2e-5, (force, pre)
                    assert np.array_equal(got, unfused), (force, pre)


@pytest.mark.parametrize("H,W,C_,N,rate,res", [(33, 65, 384, 64, 1, True), (33, 65, 576, 160, 1, False), (9, 17, 960, 160, 2, True),
                                                (5, 9, 960, 320, 2, False), (17, 33, 384, 96, 1, False), (4, 16, 64, 16, 1, True)])
def test_fused_depthwise_project(lib, H, W, C_, N, rate, res):
    """Depthwise + project in one kernel vs f64: ragged tiles (33x65 is not a multiple of 4x16), both rates, residual."""
    rng = np.random.default_rng(H + C_ + N)
    B = 2
    e = np.clip(rng.standard_normal((B, H, W, C_)) * 2 + 1, 0, 6).astype(np.float32)
    wd = (rng.standard_normal((3, 3, C_, 1)) * 0.4).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, C_).astype(np.float32), rng.standard_normal(C_).astype(np.float32)
    wp = (rng.standard_normal((C_, N)) / np.sqrt(C_)).astype(np.float32)
    sp, hp = rng.uniform(0.5, 1.5, N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((B, H, W, N)).astype(np.float32) if res else None
    y = torch.full((B, H, W, N), np.nan, device=DEV)
    panels = torch.zeros(2 * N * C_, dtype=torch.int16, device=DEV)
    rd = dev(r) if res else None
    hip.check(lib.ams_k_dw_project(PD(e), B, H, W, C_, PD(wd), rate, PD(sd), PD(hd), PD(wp), N, PD(sp), PD(hp),
                                   P(rd) if res else None, P(y), P(panels), panels.numel(), stream()))
    et = torch.as_tensor(e).double().permute(0, 3, 1, 2)
    raw = F.conv2d(F.pad(et, (rate, rate, rate, rate)), torch.as_tensor(wd).double().permute(2, 3, 0, 1), dilation=rate, groups=C_)
    d = torch.clamp(raw * torch.as_tensor(sd).view(1, -1, 1, 1) + torch.as_tensor(hd).view(1, -1, 1, 1), 0, 6).permute(0, 2, 3, 1).numpy()
    want = (d @ wp.astype(np.float64)) * sp + hp
    if res:
        want = want + r
    got = y.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) < 5e-5          # two-part bf16 split: ~1e-5 relative


# ------------------------------------------------------------------------------------------------ pooling
def test_global_mean(lib):
    rng = np.random.default_rng(0)
    B, HW, Cn = 3, 2145, 320
    x = rng.standard_normal((B, HW, Cn)).astype(np.float32) + 0.5
    y = torch.empty((B, Cn), device=DEV)
    n = lib.ams_k_global_mean_scratch(B, Cn)
    scr = torch.empty(n, device=DEV)
    hip.check(lib.ams_k_global_mean(PD(x), B, HW, Cn, P(y), P(scr), n, stream()))
    assert rel_err(y.cpu().numpy(), x.astype(np.float64).mean(axis=1)) < 1e-6


# ------------------------------------------------------------------------------------------------ head
def _np_bilinear(x, H, W):
    from oracle.student_np import resize_bilinear_align_corners
    return resize_bilinear_align_corners(x, H, W)


@pytest.mark.parametrize("h,w,H,W,cls", [(5, 9, 64, 128, [0, 1, 2, 10, 11, 13]), (3, 5, 32, 64, [2, 8, 9, 10, 11, 13]),
                                         (9, 17, 128, 256, list(range(19))), (4, 7, 50, 90, [0, 15])])
def test_upsample_argmax_metrics_and_ce_grad(lib, h, w, H, W, cls):
    rng = np.random.default_rng(h * w)
    B, NC, K = 2, 19, len(cls)
    logits = (rng.standard_normal((B, h, w, NC)) * 2).astype(np.float32)
    teacher = rng.integers(0, 19, (B, H, W)).astype(np.uint8)
    teacher[rng.random((B, H, W)) < 0.1] = 255
    ci = (C.c_int32 * K)(*cls)
    ld, td = dev(logits), torch.as_tensor(teacher).to(DEV)
    labels = torch.empty((B, H, W), dtype=torch.int32, device=DEV)
    conf = torch.empty(K * K, dtype=torch.int64, device=DEV)
    loss = torch.empty(2, dtype=torch.float64, device=DEV)
    hip.check(lib.ams_k_upsample_argmax(P(ld), B, h, w, NC, ci, K, H, W, P(td), P(labels), P(conf), P(loss), stream()))
    full = _np_bilinear(logits, H, W)[..., cls]                       # f32, same unfused lerp order as the kernel
    want_lab = np.argmax(full, axis=-1)
    assert np.array_equal(labels.cpu().numpy(), want_lab)              # bit-exact label map
    lut = np.full(256, -1)
    lut[cls] = np.arange(K)
    tgt = lut[teacher]
    valid = tgt >= 0
    cm = np.zeros((K, K), dtype=np.int64)
    np.add.at(cm, (tgt[valid], want_lab[valid]), 1)
    assert np.array_equal(conf.cpu().numpy().reshape(K, K), cm)
    z = full.astype(np.float64)
    lse = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)) + z.max(-1)
    pix = lse - np.take_along_axis(z, np.maximum(tgt, 0)[..., None], -1)[..., 0]
    got = loss.cpu().numpy()
    assert got[1] == valid.sum()
    assert got[0] / got[1] == pytest.approx(pix[valid].mean(), rel=1e-5)
    # labels only (no teacher): same labels, metric buffers untouched
    labels2 = torch.empty_like(labels)
    hip.check(lib.ams_k_upsample_argmax(P(ld), B, h, w, NC, ci, K, H, W, None, P(labels2), None, None, stream()))
    assert torch.equal(labels, labels2)
    # ---- gradient of the masked mean CE wrt the low-res logits
    from oracle.student_torch import resize_bilinear_align_corners
    lt = torch.as_tensor(logits).double().requires_grad_(True)
    zf = resize_bilinear_align_corners(lt, H, W)[..., cls]
    lse_t = torch.logsumexp(zf, -1)
    picked = torch.gather(zf, -1, torch.as_tensor(np.maximum(tgt, 0))[..., None])[..., 0]
    ((lse_t - picked)[torch.as_tensor(valid)]).mean().backward()
    dl = torch.full((B, h, w, NC), np.nan, device=DEV)
    hip.check(lib.ams_k_ce_grad(P(ld), B, h, w, NC, ci, K, H, W, P(td), P(loss), P(dl), stream()))
    assert rel_err(dl.cpu().numpy(), lt.grad.numpy()) < 2e-5
    unsel = [c for c in range(NC) if c not in cls]
    assert np.all(dl.cpu().numpy()[..., unsel] == 0)
    # ---- loss and gradient in ONE pass (what the fine-tune step runs): same loss sums, same gradient, identical bits run to run
    n = lib.ams_k_ce_loss_grad_scratch(B, h, w, K)
    scr = torch.full((n,), np.nan, device=DEV)
    outs = []
    for _ in range(2):
        loss1 = torch.full((2,), np.nan, dtype=torch.float64, device=DEV)
        dl1 = torch.full((B, h, w, NC), np.nan, device=DEV)
        hip.check(lib.ams_k_ce_loss_grad(P(ld), B, h, w, NC, ci, K, H, W, P(td), P(loss1), P(dl1), P(scr), n, stream()))
        outs.append((loss1.cpu().numpy(), dl1.cpu().numpy()))
    got1, d1 = outs[0]
    assert got1[1] == valid.sum() and got1[0] / got1[1] == pytest.approx(pix[valid].mean(), rel=1e-5)
    assert rel_err(d1, lt.grad.numpy()) < 2e-5
    assert np.all(d1[..., unsel] == 0)
    assert np.array_equal(d1, outs[1][1])


def test_upsample_all_ignored_gives_zero_count(lib):
    B, h, w, H, W, NC = 1, 3, 5, 32, 64, 19
    cls = [0, 1]
    ci = (C.c_int32 * 2)(*cls)
    logits = dev(np.random.default_rng(0).standard_normal((B, h, w, NC)))
    teacher = torch.full((B, H, W), 255, dtype=torch.uint8, device=DEV)
    labels = torch.empty((B, H, W), dtype=torch.int32, device=DEV)
    conf = torch.ones(4, dtype=torch.int64, device=DEV)
    loss = torch.ones(2, dtype=torch.float64, device=DEV)
    hip.check(lib.ams_k_upsample_argmax(P(logits), B, h, w, NC, ci, 2, H, W, P(teacher), P(labels), P(conf), P(loss), stream()))
    assert conf.sum().item() == 0 and loss.cpu().tolist() == [0.0, 0.0]


# ------------------------------------------------------------------------------------------------ Adam
def test_adam_matches_tf1_form(lib):
    from oracle.student_np import adam_step
    rng = np.random.default_rng(1)
    n = 100003
    p, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32) * 0.01
    m, v = rng.standard_normal(n).astype(np.float32) * 0.01, rng.random(n).astype(np.float32) * 1e-4
    mask = (rng.random(n) < 0.3).astype(np.uint8)
    b1p, b2p, lr = 0.9 ** 3, 0.999 ** 3, 1e-3
    lr_t = lr * np.sqrt(1 - b2p) / (1 - b1p)
    pd, gd, md, vd = dev(p), dev(g), dev(m), dev(v)
    hip.check(lib.ams_k_adam(P(pd), P(gd), P(md), P(vd), PD(mask, torch.uint8), n, float(lr_t), 0.9, 0.999, 1e-8, stream()))
    wn, mn, vn = adam_step(p.astype(np.float64), g.astype(np.float64), m.astype(np.float64), v.astype(np.float64), lr, b1p, b2p)
    assert rel_err(md.cpu().numpy(), mn) < 1e-6 and rel_err(vd.cpu().numpy(), vn) < 1e-6
    got = pd.cpu().numpy()
    assert np.array_equal(got[mask == 0], p[mask == 0])                # reverted entries keep their bits
    assert np.abs(got[mask == 1] - wn[mask == 1]).max() < 1e-6


@pytest.mark.gpu
def test_split_gemm_half_height_tail_blocks(knobs):
    """68640 x 64 -> 320 runs 160-wide tiles at two blocks per CU: 1074 full-height blocks would need 2.1 rounds, so the launcher ends the
    launch with half-height blocks (pw_plan_tail, k_pw_x3.hip).  Same products in the same order per output as the all-full plan: the
    two launches must agree bit for bit, and with a float64 product to the f32 level."""
    import ctypes as C
    import os
    lib = hip.lib()
    M, K, N = 68640, 64, 320
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) / K ** 0.5).cuda()
    sc = (torch.rand(N, generator=g) + 0.5).cuda()
    sh = torch.randn(N, generator=g).cuda()
    panels = torch.zeros(3 * N * 64, dtype=torch.int16, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        y = torch.full((M, N), float("nan"), device="cuda")
        hip.check(lib.ams_k_pointwise_split3(P(x), M, K, P(w), N, P(sc), P(sh), hip.ACT_NONE, None, P(y), P(panels), panels.numel(), st))
        torch.cuda.synchronize()
        return y

    knobs(AMS_PWX_NO_TAIL=None)
    y_tail = run()
    knobs(AMS_PWX_NO_TAIL="1")
    y_full = run()
    assert not torch.isnan(y_tail).any()
    assert torch.equal(y_tail, y_full)
    want = (x.double() @ w.double()) * sc.double() + sh.double()
    assert ((y_tail.double() - want).abs().max() / want.abs().max()).item() < 2e-6


@pytest.mark.parametrize("M,K,N,split,trans", [(17160, 960, 160, 1, 0), (17160, 160, 960, 1, 1), (4290, 576, 96, 1, 0), (2145, 64, 384, 1, 1),
                                                (1000, 384, 64, 1, 0), (33001, 96, 24, 0, 0), (70001, 192, 32, 0, 0), (40003, 32, 192, 0, 1),
                                                (300, 144, 24, 0, 0), (90, 320, 256, 0, 0),
                                                (17160, 960, 160, 2, 0), (17160, 576, 96, 2, 0), (4290, 384, 64, 2, 0), (1000, 960, 320, 2, 0)])
@pytest.mark.parametrize("x_mode", [1, 2])
def test_pointwise_with_operand_transform(lib, M, K, N, split, trans, x_mode):
    """1x1 GEMMs that apply an elementwise BN pass on their operand loads (PwArgs::x_mode; AMS_OPT_FUSE_OPERAND_BN): BN + ReLU6 of the layer
    that wrote x (mode 1), or dz = A dy + B + C z (mode 2).  Against f64 math, and BIT FOR BIT against the same kernel run on the materialised
    operand (the pass written by bn_act / bn_bwd_apply): tiled three-part kernel, streaming exact-f32 kernel (mode 1 on load, mode 2 through
    its fallback), the tiled exact-f32 kernel and ragged row counts (fallback: x' written to x_tmp first)."""
    if split == 2 and x_mode != 1:
        pytest.skip("the two-fp16-part kernel transforms FORWARD operands only (BN + activation)")
    rng = np.random.default_rng(M + K + N + x_mode)
    x = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    x2 = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K) if trans else (K, N)) / np.sqrt(K)).astype(np.float32)
    v0 = rng.uniform(0.5, 1.5, K).astype(np.float32)
    v1 = (rng.standard_normal(K) + 1).astype(np.float32)
    v2 = (rng.standard_normal(K) * 0.3).astype(np.float32)
    x64, w64 = x.astype(np.float64), (w.astype(np.float64).T if trans else w.astype(np.float64))
    if x_mode == 1:
        xp = np.clip(x64 * v0 + v1, 0.0, 6.0)
        xp32 = np.clip(x * v0 + v1, np.float32(0), np.float32(6)).astype(np.float32)          # the pass's own f32 arithmetic (mul, add, clamp)
    else:
        xp = v0.astype(np.float64) * x64 + v1 + v2.astype(np.float64) * x2
        xp32 = ((v0 * x + v1) + v2 * x2).astype(np.float32)
    want = xp @ w64
    Kp = (K + 31) // 32 * 32
    panels = torch.empty(3 * N * Kp, dtype=torch.int16, device=DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    tmp = torch.full((M, K), float("nan"), device=DEV)
    hip.check(lib.ams_k_pointwise_xform(PD(x), M, K, PD(w), N, trans, split, x_mode, hip.ACT_RELU6 if x_mode == 1 else hip.ACT_NONE, PD(v0), PD(v1),
                                        PD(v2), PD(x2), P(y), P(tmp), P(panels), panels.numel(), stream()))
    got = y.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) < (3e-5 if split else 1e-5)
    # the plain kernel on the materialised operand: same bits
    y2 = torch.empty((M, N), device=DEV)
    if split == 2:
        hip.check(lib.ams_k_pointwise_split_f16(PD(xp32), M, K, PD(w), N, None, None, hip.ACT_NONE, None, P(y2), P(panels), panels.numel(), None, None,
                                                stream()))
    elif split:
        hip.check(lib.ams_k_pointwise_split3(PD(xp32), M, K, PD(w if not trans else np.ascontiguousarray(w.T)), N, None, None, hip.ACT_NONE, None, P(y2),
                                             P(panels), panels.numel(), stream()))
    else:
        hip.check(lib.ams_k_pointwise(PD(xp32), M, K, PD(w), N, trans, None, 0, None, None, hip.ACT_NONE, None, P(y2), stream()))
    assert torch.equal(y, y2), "transform on load differs from the materialised pass by %g" % (y - y2).abs().max().item()


@pytest.mark.parametrize("M,K,N,split", [(17160, 960, 160, 1), (17160, 384, 64, 1), (4290, 576, 96, 1), (2145, 160, 960, 1), (33001, 96, 24, 0),
                                          (70001, 192, 32, 0), (513, 144, 24, 0), (90, 256, 256, 0)])
@pytest.mark.parametrize("modes", [(1, 0), (0, 2), (1, 2)])
def test_pointwise_wgrad_with_operand_transforms(lib, M, K, N, split, modes):
    """Weight gradient dw = x'^T dy' with BN + ReLU6 applied to x and / or dz = A dy + B + C z formed from (dy, z) on load (WgArgs x_mode,
    dy_mode): against f64 math and bit for bit against the same kernel on materialised operands."""
    x_mode, dy_mode = modes
    rng = np.random.default_rng(M + K + N + 7 * x_mode + dy_mode)
    x = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    z = rng.standard_normal((M, N)).astype(np.float32)
    v0 = rng.uniform(0.5, 1.5, K).astype(np.float32)
    v1 = (rng.standard_normal(K) + 1).astype(np.float32)
    d0 = rng.uniform(0.5, 1.5, N).astype(np.float32)
    d1 = (rng.standard_normal(N) * 0.5).astype(np.float32)
    d2 = (rng.standard_normal(N) * 0.3).astype(np.float32)
    xp32 = np.clip(x * v0 + v1, np.float32(0), np.float32(6)).astype(np.float32) if x_mode else x
    dp32 = ((d0 * dy + d1) + d2 * z).astype(np.float32) if dy_mode else dy
    xp = np.clip(x.astype(np.float64) * v0 + v1, 0, 6) if x_mode else x.astype(np.float64)
    dp = (d0.astype(np.float64) * dy + d1 + d2.astype(np.float64) * z) if dy_mode else dy.astype(np.float64)
    want = xp.T @ dp
    n_scr = int(lib.ams_k_pointwise_wgrad_scratch(M, K, N))
    scr = torch.empty(n_scr, device=DEV)
    dw = torch.full((K, N), float("nan"), device=DEV)
    hip.check(lib.ams_k_pointwise_wgrad_xform(PD(x), PD(dy), M, K, N, split, x_mode, hip.ACT_RELU6, PD(v0), PD(v1), dy_mode, PD(d0), PD(d1), PD(d2),
                                              PD(z), P(dw), P(scr), n_scr, stream()))
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) < 3e-5
    dw2 = torch.empty((K, N), device=DEV)
    if split:
        hip.check(lib.ams_k_pointwise_wgrad_split(PD(xp32), PD(dp32), M, K, N, P(dw2), P(scr), n_scr, stream()))
    else:
        hip.check(lib.ams_k_pointwise_wgrad(PD(xp32), PD(dp32), M, K, N, P(dw2), P(scr), n_scr, stream()))
    assert torch.equal(dw, dw2), "transforms on load differ from the materialised operands by %g" % (dw - dw2).abs().max().item()


@pytest.mark.parametrize("M,Cin,Cexp", [(265224, 24, 144), (67080, 32, 192), (1027, 16, 96), (33, 8, 32), (5000, 12, 48), (4, 32, 64)])
def test_expand_statistics_from_gram_matrix(lib, M, Cin, Cexp):
    """k_xx_stats.hip (AMS_OPT_TRAIN_RECOMPUTE = 2): XX = x^T x and g0 = sum x on the f64 matrix pipe, then the BN statistics of z = x . W per
    channel from them.  x has channel means several sigma away from zero (the cancellation inside w^T XX w is what f64 is there for); row
    counts that are not multiples of 4, fewer rows than waves, channel counts below a 16-wide chunk.  XX / g0 against NumPy f64 to 1e-12;
    scale / shift / saved statistics / moving averages against the statistics of the f64 product z to f32 rounding."""
    rng = np.random.default_rng(M + Cin + Cexp)
    x = (rng.standard_normal((M, Cin)) * rng.uniform(0.2, 2.0, Cin) + rng.standard_normal(Cin) * 3.0).astype(np.float32)
    w = (rng.standard_normal((Cin, Cexp)) / np.sqrt(Cin)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    beta = (rng.standard_normal(Cexp) * 0.1).astype(np.float32)
    center = (rng.standard_normal(Cexp) * 0.1).astype(np.float32)
    mm0 = (rng.standard_normal(Cexp) * 0.1).astype(np.float32)
    mv0 = rng.uniform(0.5, 1.5, Cexp).astype(np.float32)
    eps, omd = 1e-3, 0.1
    KP = (Cin + 15) // 16 * 16
    n_scr = int(lib.ams_k_xx_gram_scratch(M, Cin))
    scr = torch.full((n_scr,), float("nan"), dtype=torch.float64, device=DEV)
    xx64 = torch.full((KP * KP + KP,), float("nan"), dtype=torch.float64, device=DEV)
    xx32 = torch.full((KP * KP + KP,), float("nan"), device=DEV)
    hip.check(lib.ams_k_xx_gram(PD(x), M, Cin, P(scr), n_scr, P(xx64), P(xx32), stream()))
    x64 = x.astype(np.float64)
    XX = np.zeros((KP, KP)); XX[:Cin, :Cin] = x64.T @ x64
    g0 = np.zeros(KP); g0[:Cin] = x64.sum(axis=0)
    got = xx64.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got[:KP * KP].reshape(KP, KP) - XX).max() <= 1e-12 * np.abs(XX).max()
    assert np.abs(got[KP * KP:] - g0).max() <= 1e-12 * max(np.abs(g0).max(), 1.0)
    assert np.allclose(xx32.cpu().numpy(), got.astype(np.float32), rtol=1e-6, atol=0)
    scale, shift, smean, srstd = (torch.empty(Cexp, device=DEV) for _ in range(4))
    mm, mv = dev(mm0), dev(mv0)
    sums = torch.empty(2 * Cexp, dtype=torch.float64, device=DEV)
    hip.check(lib.ams_k_expand_stats(P(xx64), Cin, PD(w), Cexp, float(M), PD(center), PD(gamma), PD(beta), eps, omd, P(mm), P(mv), P(scale), P(shift),
                                     P(smean), P(srstd), P(sums), stream()))
    z = x64 @ w.astype(np.float64)
    mean, var = z.mean(axis=0), z.var(axis=0)
    rstd = 1.0 / np.sqrt(var + eps)
    assert rel_err(smean.cpu().numpy(), mean) < 1e-6 and rel_err(srstd.cpu().numpy(), rstd) < 1e-6
    assert rel_err(scale.cpu().numpy(), gamma * rstd) < 1e-6 and rel_err(shift.cpu().numpy(), beta - mean * gamma * rstd) < 2e-6
    unbiased = var * (M / max(M - 1, 1))
    assert rel_err(mm.cpu().numpy(), mm0 - (mm0 - mean) * omd) < 1e-6 and rel_err(mv.cpu().numpy(), mv0 - (mv0 - unbiased) * omd) < 1e-6
    d = z - center.astype(np.float64)
    s = sums.cpu().numpy()
    assert rel_err(s[:Cexp], d.sum(axis=0)) < 1e-9 and rel_err(s[Cexp:], (d * d).sum(axis=0)) < 1e-9

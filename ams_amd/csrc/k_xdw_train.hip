// Fine-tune step of the early inverted-residual blocks (block input <= 32 channels, blocks 1-6) WITHOUT the 6x-expanded tensors.
//
// In the layer-by-layer step the expand layer's raw output z_e, its activation a_e and the gradient da_e / dz_e are each written
// once and read two or three times: 405 MB apiece for the first such block at 8 frames of 512 x 1024, 1.4 ms of a 11.2 ms step for
// that one layer.  All three are cheap functions of the block input x (16 .. 32 channels): z_e = x . W_e is a K <= 32 product per
// pixel.  Here every consumer recomputes what it needs from x in registers (exact-f32 MFMA, the k order of the forward kernels, so the
// recomputed z_e has the forward's bits) and the expanded tensors never touch HBM:
//
//   forward   xdw_train_kernel<.., XT_FWD_STATS>   BN statistics of z_e straight from x                       (reads x)
//             expand_dw_kernel (k_expand_dw.hip)   BN + ReLU6 + depthwise 3x3 -> z_d, the inference kernel    (reads x, writes z_d)
//   backward  xdw_train_kernel<.., XT_BWD>         given dz_d = d loss / d z_d:  da_e = dwconv^T(dz_d),  dy_e = da_e . relu6'(y_e);
//                                                  BN-backward sums of the expand layer, the depthwise weight gradient
//                                                  sum a_e . dz_d, and the pieces of the expand weight gradient (below)
//                                                                                                              (reads x, dz_d)
//             xdw_dx_kernel                        dz_e = A dy_e + B + C z_e,  dx = dz_e . W_e^T (+ skip gradient)
//                                                                                                              (reads x, dz_d, writes dx)
//
// Expand weight gradient without dz_e:  dW_e[k][n] = sum_p x[p][k] dz_e[p][n]  and  dz_e = A_n dy_e + B_n + C_n z_e  per channel n, so
//   dW_e[k][n] = A_n G1[k][n] + B_n g0[k] + C_n (XX . W_e)[k][n],   G1 = x^T dy_e,  g0 = sum_p x[p],  XX = x^T x  (Cin x Cin Gram matrix)
// — all three are sums over pixels that the XT_BWD pass forms on the matrix pipe while it has x and dy_e in registers.
//
// Mapping of xdw_train_kernel: a block owns a contiguous range of 16-pixel row groups of the INPUT grid and has one wave per 16-channel
// chunk of the expanded tensor: the waves of a block read the same x (L1) and adjacent 64-byte channel slices of dz_d (together: whole
// lines), every per-chunk constant (expand weights, BN vectors, depthwise taps) lives in registers for the wave's lifetime, and every
// reduction over pixels stays in registers until the wave is done — one partial row per block, summed afterwards in a fixed order
// (deterministic).  At stride 2 a row group holds pixels of ONE parity class of (row + pad, column + pad): the taps that reach a pixel
// depend on that parity only (4 / 2 / 2 / 1 of the 9), so they are wave-uniform and statically indexed.
// xdw_dx_kernel: a wave owns row groups and walks the chunks (the reduction there runs over the expanded channels).
#include <type_traits>

#include "pw_common.hpp"

namespace ams {

enum { XT_FWD_STATS = 0, XT_BWD = 1 };

struct XtArgs {
    const float* x;            // [B, H, W, Cin]  block input
    int B, H, W, Cin;
    const float* w_exp;        // [Cin, Cexp]
    int Cexp;
    const float* center;       // FWD_STATS: per-channel centre of the shifted sums (moving mean), may be null
    const float* sc_e; const float* sh_e; const float* mean_e; const float* rstd_e;     // BWD: training BN of the expand layer
    int act_e;
    const float* w_dw;         // [9, Cexp]
    const float* dz_d;         // [B, Ho, Wo, Cexp]  gradient wrt the depthwise layer's raw output
    int Ho, Wo, pt, pl;
    // dx pass
    const float* cA; const float* cB; const float* cC;       // dz_e = cA dy_e + cB + cC z_e
    const float* res;          // gradient arriving over the skip connection [B, H, W, Cin], or null
    float* dx;                 // [B, H, W, Cin]
    // dx pass, optional: dx is the gradient wrt the output of a BN layer WITHOUT activation (the previous block's project layer) whose raw
    // output is red_z: sum(dx) and sum(dx xhat) come out as one partial row [2][Cin] per block (first half of that layer's BN backward)
    const float* red_z; const float* red_mean; const float* red_rstd;
    float* red_part;
    // partial rows (one per block):  S [2][Cexp] | dWd [9][Cexp] | G1 [KP][Cexp] | XX [KP][KP] | g0 [KP]   (KP = 16 KC)
    float* part;
    int64_t part_stride;       // floats per block row
    int tiles_y, tiles_x, n_tiles;      // tiles of 64 input pixels per image: 4 x 16 (stride 1) / 8 x 8 in padded coordinates (stride 2)
    // STEM form (first block of the network): the "expand" layer is the stem conv = a 1x1 conv over the 27-tap patch of the normalised,
    // 127.5-padded frame (k = tap * 3 + channel), gathered per pixel from the frames [B, fH, fW, 3]; H x W is the stem's output grid
    const void* frames;
    int fH, fW, spt, spl;
    float ps;
};

// value of the normalised, 127.5-padded frame at (iy, ix, ch) in padded coordinates; outside of it the stem's SAME zero padding
// (two roundings, x * ps - 1, as the graph's Mul and Sub)
template <typename TIn>
__device__ __forceinline__ float xt_frame_value(const TIn* img, int H, int W, int iy, int ix, int ch, float ps) {
    const bool inside = iy >= 0 && ix >= 0 && iy <= H && ix <= W;
    const bool pad = iy >= H || ix >= W;
    const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const int ixc = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    float raw = (float)img[(iyc * W + ixc) * 3 + ch];
    raw = pad ? 127.5f : raw;
    const float v = __fsub_rn(__fmul_rn(raw, ps), 1.0f);
    return inside ? v : 0.f;
}

// Tile geometry.  A tile is 64 INPUT pixels = four 16-pixel row groups:
//   stride 1: 4 rows x 16 columns, row group = tile row; the gradient values it needs are a 6 x 18 patch of dz_d;
//   stride 2: 8 x 8 in padded coordinates (n = i + pad), row group = parity class (py, px) of (ny, nx), lane l15 = (a, b) with
//             ny = ny0 + 2a + py: the taps that reach a pixel depend on its class only (4 / 2 / 2 / 1 of the 9) and come from a
//             5 x 5 patch of dz_d.
// The patch (16 channels of it) is staged in the wave's LDS, zero outside the map, so a tap is one ds_read_b128 at a constant offset
// from the lane's base position: no per-tap address arithmetic, no border masks.
template <int S>
struct XtGeo {
    static constexpr int TPH = S == 1 ? 6 : 5, TPW = S == 1 ? 18 : 5, NPOS = TPH * TPW;
    static constexpr int PITCH = 20;                                   // floats per patch position (16 channels + pad)
    static constexpr int NU = (NPOS * 4 + 63) / 64;                    // float4 loads per lane to stage a patch
    static constexpr int TAP_FLOATS = NPOS * PITCH;
    // taps (i, j) valid for row group rg: stride 1 all nine; stride 2 by parity
    static __device__ __forceinline__ constexpr bool tap_ok(int rg, int i, int j) {
        return S == 1 ? true : (((rg >> 1) == 0 ? (i != 1) : (i == 1)) && ((rg & 1) == 0 ? (j != 1) : (j == 1)));
    }
    // patch position of tap (i, j) relative to the lane's base position
    static __device__ __forceinline__ constexpr int tap_off(int i, int j) { return S == 1 ? -(i * TPW + j) : -((i >> 1) * TPW + (j >> 1)); }
};

struct XtTileCtx { int b, oy0, ox0, y0, x0; };        // image, patch origin in dz_d, tile origin (stride 1: input; stride 2: padded coordinates)

template <int S>
__device__ __forceinline__ XtTileCtx xt_tile(const XtArgs& a, int t) {
    XtTileCtx c;
    const int tx = t % a.tiles_x, t2 = t / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    c.b = t2 / a.tiles_y;
    if (S == 1) { c.y0 = 4 * ty; c.x0 = 16 * tx; c.oy0 = c.y0 + a.pt - 2; c.ox0 = c.x0 + a.pl - 2; }
    else { c.y0 = 8 * ty; c.x0 = 8 * tx; c.oy0 = c.y0 / 2 - 1; c.ox0 = c.x0 / 2 - 1; }
    return c;
}

// input pixel of (row group rg, lane l15); returns whether it exists, clamped coordinates either way
template <int S>
__device__ __forceinline__ bool xt_pixel(const XtArgs& a, const XtTileCtx& c, int rg, int l15, int& iy, int& ix) {
    if (S == 1) { iy = c.y0 + rg; ix = c.x0 + l15; }
    else { iy = c.y0 + 2 * (l15 >> 2) + (rg >> 1) - a.pt; ix = c.x0 + 2 * (l15 & 3) + (rg & 1) - a.pl; }
    const bool live = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    iy = iy < 0 ? 0 : (iy > a.H - 1 ? a.H - 1 : iy);
    ix = ix < 0 ? 0 : (ix > a.W - 1 ? a.W - 1 : ix);
    return live;
}

// lane's base position in the patch for row group rg (tap (i, j) sits at base + tap_off(i, j))
template <int S>
__device__ __forceinline__ int xt_base(int rg, int l15) {
    return S == 1 ? (rg + 2) * XtGeo<S>::TPW + l15 + 2 : ((l15 >> 2) + 1) * XtGeo<S>::TPW + (l15 & 3) + 1;
}

// 16 channels (n0 ..) of the tile's dz_d patch: global -> registers (zero outside the map; clamped addresses, no branches)
template <int S>
__device__ __forceinline__ void xt_patch_load(const XtArgs& a, const XtTileCtx& c, int n0, int lane, float4 (&v)[XtGeo<S>::NU]) {
    typedef XtGeo<S> G;
    // The lane's patch positions do not depend on the tile: left visible, the compiler keeps (row, column, 64-bit offset) of all NU of
    // them in registers across the whole kernel (28+ VGPRs at stride 1: the kernel spilled).  Opaque lane id -> recomputed per tile
    // (a few VALU per load), 32-bit element offsets from a wave-uniform base (one image of dz_d is < 2^31 elements: checked on the host).
    asm volatile("" : "+v"(lane));
    const float* base = a.dz_d + (int64_t)c.b * a.Ho * a.Wo * a.Cexp + n0;
#pragma unroll
    for (int u = 0; u < G::NU; ++u) {
        int e = lane + 64 * u;
        if (e > G::NPOS * 4 - 1) e = G::NPOS * 4 - 1;
        const int pos = e >> 2, quad = e & 3;
        const int r = pos / G::TPW, cc = pos - r * G::TPW;
        const int oy = c.oy0 + r, ox = c.ox0 + cc;
        const bool ok = oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
        const int oyc = oy < 0 ? 0 : (oy > a.Ho - 1 ? a.Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox > a.Wo - 1 ? a.Wo - 1 : ox);
        const float4 t = ld4(base + (oyc * a.Wo + oxc) * a.Cexp + 4 * quad);
        v[u] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
}
template <int S>
__device__ __forceinline__ void xt_patch_store(float* sTap, int lane, const float4 (&v)[XtGeo<S>::NU]) {
    typedef XtGeo<S> G;
#pragma unroll
    for (int u = 0; u < G::NU; ++u) {
        const int e = lane + 64 * u;
        if (e < G::NPOS * 4) st4(sTap + (e >> 2) * G::PITCH + 4 * (e & 3), v[u]);
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float4 sum16(float4 v) {                   // over the 16 pixels of a row group (lanes l15), per q group
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
        v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    return v;
}

template <int MODE, int S, int KC, typename TIn = void>
__global__ __launch_bounds__(std::is_void<TIn>::value ? 768 : 128) void xdw_train_kernel(XtArgs a) {
    constexpr bool STEM = !std::is_void<TIn>::value;      // the block input is gathered from the frames; XX / g0 are formed here (no forward pass did)
    typedef XtGeo<S> G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = 16 * wave;
    // BWD: per wave the dz_d patch [NPOS][PITCH] + transposition tiles x [KC][16][20], dy [16][20]; then the block's table
    // sc_e | sh_e | mean_e | rstd_e | taps [9][Cexp].  FWD_STATS: one x transposition tile (wave 0 forms the Gram matrix of x).
    constexpr int WAVE_LDS = MODE == XT_BWD ? G::TAP_FLOATS + (KC + 1) * 16 * 20 : 0;
    float* sTap = smem + wave * WAVE_LDS;
    float* sX = MODE == XT_BWD ? sTap + G::TAP_FLOATS : smem;
    float* sD = sX + KC * 16 * 20;
    float* sVec = smem + (blockDim.x >> 6) * WAVE_LDS;
    const float* sW9 = sVec + 4 * a.Cexp;
    if (MODE == XT_BWD) {
        for (int e = tid; e < 13 * a.Cexp; e += blockDim.x) {
            const int which = e / a.Cexp, c = e - which * a.Cexp;
            const float* src = which == 0 ? a.sc_e : which == 1 ? a.sh_e : which == 2 ? a.mean_e : which == 3 ? a.rstd_e : a.w_dw + (which - 4) * a.Cexp;
            sVec[e] = src[c];
        }
        __syncthreads();
    }
    // this wave's constants: expand weights (MFMA operand A: w[k = 16c + 4q + j][n0 + l15]) and the BN vectors of channels n0 + 4q ..
    float wa[KC][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * c + 4 * q + j;
            wa[c][j] = k < a.Cin ? a.w_exp[(int64_t)k * a.Cexp + n0 + l15] : 0.f;
        }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ctr = z4;
    if (MODE == XT_FWD_STATS && a.center) ctr = ld4(a.center + n0 + 4 * q);
    const float lo = a.act_e == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = a.act_e == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    float4 s1 = z4, s2 = z4, dwd[9], g0[KC];
    f32x4 g1[KC], xx[KC][KC];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwd[k] = z4;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        g0[c] = z4;
        g1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c2 = 0; c2 < KC; ++c2) xx[c][c2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the block's contiguous share of the tiles
    const int per = (a.n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_lo = blockIdx.x * per, t_hi = t_lo + per < a.n_tiles ? t_lo + per : a.n_tiles;
    for (int t = t_lo; t < t_hi; ++t) {
        const XtTileCtx tc = xt_tile<S>(a, t);
        // ---- every global load of the tile is requested up front: the block input of its 64 pixels and the dz_d patch
        constexpr bool ROLLED = S == 1 && MODE == XT_BWD;              // see `group` below
        auto load_x = [&](int rg, float4 (&xv)[KC], float& m) {
            int iy, ix;
            const bool live = xt_pixel<S>(a, tc, rg, l15, iy, ix);
            m = live ? 1.f : 0.f;
            if constexpr (STEM) {
                // the 27-tap patch of the normalised frame under this stem pixel, k = tap * 3 + channel (the im2col row).  Branch-free
                // border form throughout: an interior fast path (plain loads at constant offsets) measured the same 195 us — the pass
                // is not bound by this gather
                typedef typename std::conditional<STEM, TIn, float>::type TI;
                const TI* img = reinterpret_cast<const TI*>(a.frames) + (int64_t)tc.b * a.fH * a.fW * 3;
                const int fy = iy * 2 - a.spt, fx = ix * 2 - a.spl;
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    float t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * c + 4 * q + j, tap = k / 3;
                        const int ch = k - tap * 3, dy = tap / 3, dx = tap - dy * 3;
                        const float v = xt_frame_value(img, a.fH, a.fW, fy + (k < 27 ? dy : 0), fx + (k < 27 ? dx : 0), k < 27 ? ch : 0, a.ps);
                        t[j] = (k < 27 && live) ? v : 0.f;
                    }
                    xv[c] = make_float4(t[0], t[1], t[2], t[3]);
                }
                return;
            }
            const float* px = a.x + (int64_t)tc.b * a.H * a.W * a.Cin + (iy * a.W + ix) * a.Cin;      // wave-uniform base + 32-bit offset
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                int koff = 16 * c + 4 * q;
                const bool ok = koff < a.Cin && live;
                if (koff > a.Cin - 4) koff = a.Cin - 4;
                const float4 v = ld4(px + koff);
                xv[c] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        };
        float4 x4[ROLLED ? 1 : 4][KC];
        float lv[ROLLED ? 1 : 4];
#pragma unroll
        for (int rg = 0; rg < (ROLLED ? 1 : 4); ++rg) load_x(rg, x4[rg], lv[rg]);
        if (MODE == XT_BWD) {
            float4 pv[G::NU];
            xt_patch_load<S>(a, tc, n0, lane, pv);
            xt_patch_store<S>(sTap, lane, pv);
            wave_lds_fence();
        }
        // one 16-pixel row group: z_e -> BN -> taps -> sums.  RGC >= 0: compile-time row group (stride 2: the tap set depends on it);
        // RGC < 0: run-time row group (stride 1: all nine taps for every group; the loop over the groups stays rolled, which keeps the
        // 36 depthwise-gradient accumulators in registers — fully unrolled, hipcc interleaved the four groups and spilled them)
        auto group = [&](auto rgc, int rg, const float4 (&xv)[KC], float m) {
            constexpr int RGC = decltype(rgc)::value;
            // z_e for (pixel l15, channels n0 + 4q ..): weights = MFMA operand A, the k order of pw_gemm_f32_s / expand_dw_kernel
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][0], xv[c].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][1], xv[c].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][2], xv[c].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][3], xv[c].w, acc, 0, 0, 0);
            }
            const float4 z = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (MODE == XT_FWD_STATS) {
                float4 d = sub4_pk(z, ctr);
                d = make_float4(d.x * m, d.y * m, d.z * m, d.w * m);      // m = 0 for a position of the tile outside the image
                s1 = add4_pk(s1, d);
                s2 = add4_pk(s2, mul4_pk(d, d));
                if (wave == 0) {
                    // the sums over pixels of x and of x x^T (Cin x Cin): the backward pass needs them for the expand weight gradient
                    // (file header) and x is the same then; contraction over PIXELS -> x goes through an LDS tile to swap lanes and registers
#pragma unroll
                    for (int c = 0; c < KC; ++c) { st4(sX + (c * 16 + l15) * 20 + 4 * q, xv[c]); g0[c] = add4_pk(g0[c], xv[c]); }
                    wave_lds_fence();
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float xt[KC];
#pragma unroll
                        for (int c = 0; c < KC; ++c) xt[c] = sX[(c * 16 + 4 * gq + q) * 20 + l15];
#pragma unroll
                        for (int c = 0; c < KC; ++c)
#pragma unroll
                            for (int c2 = 0; c2 < KC; ++c2) xx[c][c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[c], xt[c2], xx[c][c2], 0, 0, 0);
                    }
                    wave_lds_fence();
                }
                return;
            }
            const float4 sc = ld4(sVec + n0 + 4 * q), sh = ld4(sVec + a.Cexp + n0 + 4 * q);
            const float4 y = muladd4_pk(z, sc, sh);                   // two roundings, as bn_act / the forward kernels
            const float4 ae = make_float4(__builtin_amdgcn_fmed3f(y.x, lo, hi) * m, __builtin_amdgcn_fmed3f(y.y, lo, hi) * m,
                                          __builtin_amdgcn_fmed3f(y.z, lo, hi) * m, __builtin_amdgcn_fmed3f(y.w, lo, hi) * m);
            float4 da = z4;
            const float* tb = sTap + xt_base<S>(rg, l15) * G::PITCH + 4 * q;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (G::tap_ok(RGC < 0 ? 0 : RGC, i, j)) {
                        const float4 gv = ld4(tb + G::tap_off(i, j) * G::PITCH);
                        fma4_pk(da, gv, ld4(sW9 + (i * 3 + j) * a.Cexp + n0 + 4 * q));
                        fma4_pk(dwd[i * 3 + j], ae, gv);               // depthwise weight gradient: a_e here x dz_d at the tap
                    }
            // Relu6Grad: strictly inside (0, 6); nothing from positions outside the image
            const float4 dy = make_float4((y.x > lo && y.x < hi) ? da.x * m : 0.f, (y.y > lo && y.y < hi) ? da.y * m : 0.f,
                                          (y.z > lo && y.z < hi) ? da.z * m : 0.f, (y.w > lo && y.w < hi) ? da.w * m : 0.f);
            s1 = add4_pk(s1, dy);
            {
                const float4 mu = ld4(sVec + 2 * a.Cexp + n0 + 4 * q), rs = ld4(sVec + 3 * a.Cexp + n0 + 4 * q);
                s2 = add4_pk(s2, mul4_pk(mul4_pk(dy, sub4_pk(z, mu)), rs));
            }
            // pieces of the expand weight gradient: contraction over PIXELS, so x and dy_e go through the wave's LDS tile to swap the
            // roles of lanes and registers (lane (l15, q) then holds pixel 4g + q, channel l15)
#pragma unroll
            for (int c = 0; c < KC; ++c) st4(sX + (c * 16 + l15) * 20 + 4 * q, xv[c]);
            st4(sD + l15 * 20 + 4 * q, dy);
            if (STEM && wave == 0) {
#pragma unroll
                for (int c = 0; c < KC; ++c) g0[c] = add4_pk(g0[c], xv[c]);
            }
            wave_lds_fence();
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float xt[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) xt[c] = sX[(c * 16 + 4 * gq + q) * 20 + l15];
                const float dt = sD[(4 * gq + q) * 20 + l15];
#pragma unroll
                for (int c = 0; c < KC; ++c) g1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[c], dt, g1[c], 0, 0, 0);
                if (STEM && wave == 0) {                  // no forward statistics pass ran over the patches: their Gram matrix is formed here
#pragma unroll
                    for (int c = 0; c < KC; ++c)
#pragma unroll
                        for (int c2 = 0; c2 < KC; ++c2) xx[c][c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[c], xt[c2], xx[c][c2], 0, 0, 0);
                }
            }
            wave_lds_fence();
        };
        if constexpr (ROLLED) {
            float4 xc[KC];
            float mc = lv[0];
#pragma unroll
            for (int c = 0; c < KC; ++c) xc[c] = x4[0][c];
#pragma unroll 1
            for (int rg = 0; rg < 4; ++rg) {
                float4 xn[KC];
                float mn;
                load_x(rg < 3 ? rg + 1 : 3, xn, mn);                  // the next group's input: in flight across this group's work
                group(std::integral_constant<int, -1>{}, rg, xc, mc);
#pragma unroll
                for (int c = 0; c < KC; ++c) xc[c] = xn[c];
                mc = mn;
            }
        } else {
            group(std::integral_constant<int, 0>{}, 0, x4[0], lv[0]);
            group(std::integral_constant<int, 1>{}, 1, x4[1], lv[1]);
            group(std::integral_constant<int, 2>{}, 2, x4[2], lv[2]);
            group(std::integral_constant<int, 3>{}, 3, x4[3], lv[3]);
        }
    }
    // ---- the wave's partial sums -> the block's partial row (every element written by exactly one lane)
    float* row = a.part + (int64_t)blockIdx.x * a.part_stride;
    constexpr int KP = 16 * KC;
    s1 = sum16(s1); s2 = sum16(s2);
    if (l15 == 0) { st4(row + n0 + 4 * q, s1); st4(row + a.Cexp + n0 + 4 * q, s2); }
    if (MODE == XT_FWD_STATS) {
        if (wave == 0) {                                // XX [KP][KP] | g0 [KP] behind the statistics
            float* rX = row + 2 * (int64_t)a.Cexp;
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int c2 = 0; c2 < KC; ++c2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rX[(16 * c + 4 * q + r) * KP + 16 * c2 + l15] = xx[c][c2][r];
            float* r0 = rX + KP * KP;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const float4 t = sum16(g0[c]);
                if (l15 == 0) st4(r0 + 16 * c + 4 * q, t);
            }
        }
        return;
    }
    float* rD = row + 2 * (int64_t)a.Cexp;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 t = sum16(dwd[k]);
        if (l15 == 0) st4(rD + (int64_t)k * a.Cexp + n0 + 4 * q, t);
    }
    float* rG = rD + 9 * (int64_t)a.Cexp;
    // G1 accumulator c: rows k = 16c + 4q + r, column n0 + l15
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) rG[(int64_t)(16 * c + 4 * q + r) * a.Cexp + n0 + l15] = g1[c][r];
    if (STEM && wave == 0) {                            // XX [KP][KP] | g0 [KP] behind G1
        float* rX = rG + (int64_t)KP * a.Cexp;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int c2 = 0; c2 < KC; ++c2)
#pragma unroll
                for (int r = 0; r < 4; ++r) rX[(16 * c + 4 * q + r) * KP + 16 * c2 + l15] = xx[c][c2][r];
        float* r0 = rX + KP * KP;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const float4 t = sum16(g0[c]);
            if (l15 == 0) st4(r0 + 16 * c + 4 * q, t);
        }
    }
}

// ---- pass 2: dx.  A wave owns tiles and walks the 16-channel chunks: the patch of the next chunk is in flight (registers) while the
// current one is consumed from LDS.
// Registers: the stride-1 form with two k chunks (Cin 17 .. 32: blocks 2, 4, 5) took 256 + 48 of them = ONE wave per SIMD for a kernel that waits on
// its patch loads; the per-channel constants of the fused reduction and the pixels' offsets are re-formed in the epilogue instead of living across
// the chunk loop, and the bound of two blocks per CU holds the rest under 256 (two waves per SIMD).
template <int S, int KC>
__global__ __launch_bounds__(256, KC == 1 ? 3 : 2) void xdw_dx_kernel(XtArgs a) {
    typedef XtGeo<S> G;
    constexpr int NTO = KC;                              // 16-wide tiles of dx (Cin <= 16 KC)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sVec = smem;                                  // sc_e | sh_e | cA | cB | cC | taps [9][Cexp]
    for (int e = threadIdx.x; e < 14 * a.Cexp; e += 256) {
        const int which = e / a.Cexp, c = e - which * a.Cexp;
        const float* src = which == 0 ? a.sc_e : which == 1 ? a.sh_e : which == 2 ? a.cA : which == 3 ? a.cB : which == 4 ? a.cC
                                                                                                             : a.w_dw + (which - 5) * a.Cexp;
        sVec[e] = src[c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sTap = smem + 14 * a.Cexp + wave * G::TAP_FLOATS;
    const float lo = a.act_e == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = a.act_e == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    const int chunks = a.Cexp / 16;
    const bool red = a.red_part != nullptr;              // block-uniform
    float4 rs1[NTO], rs2[NTO];
#pragma unroll
    for (int tt = 0; tt < NTO; ++tt) { rs1[tt] = make_float4(0.f, 0.f, 0.f, 0.f); rs2[tt] = rs1[tt]; }
    for (int t = blockIdx.x * 4 + wave; t < a.n_tiles; t += gridDim.x * 4) {
        const XtTileCtx tc = xt_tile<S>(a, t);
        float4 x4[4][KC];
        const int64_t img_off = (int64_t)tc.b * a.H * a.W * a.Cin;      // wave-uniform; a pixel's offset inside its image fits 32 bits (checked on the host)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            int iy, ix;
            xt_pixel<S>(a, tc, rg, l15, iy, ix);
            const float* px = a.x + img_off + (iy * a.W + ix) * a.Cin;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                int koff = 16 * c + 4 * q;
                const bool ok = koff < a.Cin;
                if (koff > a.Cin - 4) koff = a.Cin - 4;
                const float4 v = ld4(px + koff);
                x4[rg][c] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        }
        f32x4 out[4][NTO];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int tt = 0; tt < NTO; ++tt) out[rg][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 pv[G::NU];
        xt_patch_load<S>(a, tc, 0, lane, pv);
        for (int ci = 0; ci < chunks; ++ci) {
            const int n0 = ci * 16;
            xt_patch_store<S>(sTap, lane, pv);
            wave_lds_fence();
            xt_patch_load<S>(a, tc, ci + 1 < chunks ? n0 + 16 : n0, lane, pv);      // next chunk's patch: in flight across the compute
            // chunk constants: expand weights in both orientations (L1 / L2), BN and coefficient vectors from the block's LDS table
            float wa[KC][4], wp[4][NTO];
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * c + 4 * q + j;
                    const float w = a.w_exp[(int64_t)(k < a.Cin ? k : a.Cin - 1) * a.Cexp + n0 + l15];
                    wa[c][j] = k < a.Cin ? w : 0.f;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tt = 0; tt < NTO; ++tt) {
                    const int k = 16 * tt + l15;
                    const float w = a.w_exp[(int64_t)(k < a.Cin ? k : a.Cin - 1) * a.Cexp + n0 + 4 * q + j];
                    wp[j][tt] = k < a.Cin ? w : 0.f;
                }
            const float* vv = sVec + n0 + 4 * q;
            const float4 sc = ld4(vv), sh = ld4(vv + a.Cexp), cA = ld4(vv + 2 * a.Cexp), cB = ld4(vv + 3 * a.Cexp), cC = ld4(vv + 4 * a.Cexp);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][0], x4[rg][c].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][1], x4[rg][c].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][2], x4[rg][c].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][3], x4[rg][c].w, acc, 0, 0, 0);
                }
                const float4 z = make_float4(acc[0], acc[1], acc[2], acc[3]);
                const float4 y = muladd4_pk(z, sc, sh);
                float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* tb = sTap + xt_base<S>(rg, l15) * G::PITCH + 4 * q;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (G::tap_ok(rg, i, j)) fma4_pk(da, ld4(tb + G::tap_off(i, j) * G::PITCH), ld4(vv + (5 + i * 3 + j) * a.Cexp));
                const float4 dy = make_float4((y.x > lo && y.x < hi) ? da.x : 0.f, (y.y > lo && y.y < hi) ? da.y : 0.f,
                                              (y.z > lo && y.z < hi) ? da.z : 0.f, (y.w > lo && y.w < hi) ? da.w : 0.f);
                // dz_e = A dy + B + C z, evaluated left to right as bn_bwd_apply does
                const float4 dz = add4_pk(add4_pk(mul4_pk(cA, dy), cB), mul4_pk(cC, z));
                const float dv[4] = {dz.x, dz.y, dz.z, dz.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tt = 0; tt < NTO; ++tt) out[rg][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[j][tt], dv[j], out[rg][tt], 0, 0, 0);
            }
            wave_lds_fence();                            // the patch is consumed: the next chunk's may overwrite it
        }
        // epilogue: + the gradient over the skip connection, 16-byte stores (lane: pixel l15, channels 16t + 4q ..)
#pragma unroll
        for (int tt = 0; tt < NTO; ++tt) {
            const int c4 = 16 * tt + 4 * q;
            if (c4 >= a.Cin) continue;
            float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rsd = mu;
            if (red) { mu = ld4(a.red_mean + c4); rsd = ld4(a.red_rstd + c4); }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                int iy, ix;
                if (!xt_pixel<S>(a, tc, rg, l15, iy, ix)) continue;
                const int64_t off = img_off + (iy * a.W + ix) * a.Cin + c4;
                float4 v = make_float4(out[rg][tt][0], out[rg][tt][1], out[rg][tt][2], out[rg][tt][3]);
                if (a.res) v = add4_pk(v, ld4(a.res + off));
                st4(a.dx + off, v);
                if (red) {
                    const float4 zp = ld4(a.red_z + off);
                    rs1[tt] = add4_pk(rs1[tt], v);
                    rs2[tt] = add4_pk(rs2[tt], mul4_pk(mul4_pk(v, sub4_pk(zp, mu)), rsd));
                }
            }
        }
    }
    if (red) {
        // lanes of a wave that share q hold the same channels: add the 16 pixels, then the four waves in a fixed order; one row per block
        __syncthreads();                                 // every wave is done with the LDS table and its tap patch
        float* sRed = smem;                              // [4 waves][2][16 NTO]
#pragma unroll
        for (int tt = 0; tt < NTO; ++tt) {
            const float4 u = sum16(rs1[tt]), v = sum16(rs2[tt]);
            if (l15 == 0) {
                st4(sRed + (wave * 2 + 0) * 16 * NTO + 16 * tt + 4 * q, u);
                st4(sRed + (wave * 2 + 1) * 16 * NTO + 16 * tt + 4 * q, v);
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * 16 * NTO; e += 256) {
            const int which = e / (16 * NTO), c = e - which * (16 * NTO);
            if (c >= a.Cin) continue;
            float s = 0.f;
            for (int wv = 0; wv < 4; ++wv) s += sRed[(wv * 2 + which) * 16 * NTO + c];
            a.red_part[(int64_t)blockIdx.x * 2 * a.Cin + which * a.Cin + c] = s;
        }
    }
}

// dW_e[k][n] = A_n G1[k][n] + B_n g0[k] + C_n sum_k' XX[k][k'] W_e[k'][n]   (sums = the reduced partial row: G1 | XX | g0)
__global__ void xdw_dwe_kernel(const float* __restrict__ G1, const float* __restrict__ XX, int KP, int Cin, int Cexp,
                               const float* __restrict__ w_exp, const float* __restrict__ cA, const float* __restrict__ cB,
                               const float* __restrict__ cC, float* __restrict__ dw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cin * Cexp) return;
    const int k = e / Cexp, n = e - k * Cexp;
    const float* g0 = XX + KP * KP;
    double xw = 0.0;
    for (int k2 = 0; k2 < Cin; ++k2) xw += (double)XX[k * KP + k2] * (double)w_exp[(int64_t)k2 * Cexp + n];
    dw[e] = (float)((double)cA[n] * (double)G1[(int64_t)k * Cexp + n] + (double)cB[n] * (double)g0[k] + (double)cC[n] * xw);
}

// ---- host side -----------------------------------------------------------------------------------------------------
bool xdw_train_supported(int Cin, int Cexp, int stride, int rate) {
    return Cin % 4 == 0 && Cin >= 8 && Cin <= 32 && rate == 1 && (stride == 1 || stride == 2) && Cexp % 16 == 0 && Cexp >= 32 && Cexp <= 192;
}

static void xt_geometry(XtArgs& a, int stride) {
    same_pad(a.H, 3, stride, 1, &a.Ho, &a.pt);
    same_pad(a.W, 3, stride, 1, &a.Wo, &a.pl);
    if (stride == 1) { a.tiles_y = cdiv(a.H, 4); a.tiles_x = cdiv(a.W, 16); }
    else { a.tiles_y = cdiv(a.H + a.pt, 8); a.tiles_x = cdiv(a.W + a.pl, 8); }      // 8 x 8 in padded coordinates
    a.n_tiles = a.B * a.tiles_y * a.tiles_x;
}

static int64_t xt_part_stride(int KP, int Cexp) { return ((int64_t)(2 + 9 + KP) * Cexp + 3) / 4 * 4; }       // backward: S | dWd | G1
static int64_t xt_fwd_stride(int KP, int Cexp) { return ((int64_t)2 * Cexp + KP * KP + KP + 3) / 4 * 4; }     // forward: S | XX | g0

// blocks (= partial rows) of the passes over a block with Cexp expanded channels.  A block has Cexp / 16 waves; the stem (2 waves per
// block) needs 2048 blocks to fill the chip (328 -> 194 us); for 6 / 9 waves 1024 / 682 blocks measured 15-20 % slower than 512
int xdw_train_blocks(int B, int H, int W, int Cexp) {
    const int64_t tiles = cdiv64((int64_t)B * H * W, 64);
    int64_t blocks = Cexp / 16 <= 2 ? 2048 : 512;
    if (blocks > tiles / 4) blocks = tiles / 4;        // >= 4 tiles (256 pixels) per wave
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

size_t xdw_train_scratch(int B, int H, int W, int Cin, int Cexp) {
    if ((int64_t)H * W * (Cexp > Cin ? Cexp : Cin) >= 0x7fffffffLL) return (size_t)-1;      // the kernels use 32-bit element offsets inside an image: such a map takes the layer-by-layer step
    const int KP = (Cin + 15) / 16 * 16;
    // partial rows + the reduced row
    const int64_t st = xt_part_stride(KP, Cexp) > xt_fwd_stride(KP, Cexp) ? xt_part_stride(KP, Cexp) : xt_fwd_stride(KP, Cexp);
    return (size_t)(xdw_train_blocks(B, H, W, Cexp) + 1) * st;
}

template <int MODE, int S>
static int launch_xt(const XtArgs& a, int blocks, hipStream_t st) {
    const int KC = (a.Cin + 15) / 16;
    const int waves = a.Cexp / 16;
    const size_t lds = (MODE == XT_BWD ? (size_t)waves * (XtGeo<S>::TAP_FLOATS + (KC + 1) * 16 * 20) + 13 * a.Cexp : (size_t)KC * 16 * 20) * sizeof(float);
    AMS_REQUIRE(lds <= 160 * 1024 - 512, "xdw_train: %zu bytes of LDS", lds);
    if (blocks > a.n_tiles) blocks = a.n_tiles;
    note_kernel(MODE == XT_BWD ? "xdw_train_kernel<BWD>" : "xdw_train_kernel<FWD_STATS>");
    if (KC == 1) {
        RUN_RC(func_allow_lds((const void*)xdw_train_kernel<MODE, S, 1>, lds));
        hipLaunchKernelGGL((xdw_train_kernel<MODE, S, 1>), dim3(blocks), dim3(64 * waves), lds, st, a);
    } else {
        RUN_RC(func_allow_lds((const void*)xdw_train_kernel<MODE, S, 2>, lds));
        hipLaunchKernelGGL((xdw_train_kernel<MODE, S, 2>), dim3(blocks), dim3(64 * waves), lds, st, a);
    }
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// forward: per-channel shifted sums of z_e = x . W_e over all pixels, and the sums of x and x x^T -> partial rows
// S [2][Cexp] | XX [KP][KP] | g0 [KP] in scratch, *stride_out floats apart
int launch_xdw_fwd_stats(const float* x, int B, int H, int W, int Cin, const float* w_exp, int Cexp, const float* center, float* scratch,
                         int* rows_out, int64_t* stride_out, hipStream_t st) {
    AMS_REQUIRE(xdw_train_supported(Cin, Cexp, 1, 1), "xdw_fwd_stats: unsupported shape Cin=%d Cexp=%d", Cin, Cexp);
    XtArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.w_exp = w_exp; a.Cexp = Cexp; a.center = center;
    xt_geometry(a, 1);
    int blocks = xdw_train_blocks(B, H, W, Cexp);
    if (blocks > a.n_tiles) blocks = a.n_tiles;
    a.part = scratch; a.part_stride = xt_fwd_stride((Cin + 15) / 16 * 16, Cexp);
    *rows_out = blocks; *stride_out = a.part_stride;
    return launch_xt<XT_FWD_STATS, 1>(a, blocks, st);
}

// backward pass 1: partial rows (S | dWd | G1 | XX | g0) in scratch; *rows_out rows of *stride_out floats
int launch_xdw_bwd_reduce(const float* x, int B, int H, int W, int Cin, const float* w_exp, int Cexp, const float* sc_e, const float* sh_e,
                          const float* mean_e, const float* rstd_e, int act_e, const float* w_dw, int stride, const float* dz_d, float* scratch,
                          int* rows_out, int64_t* stride_out, hipStream_t st) {
    AMS_REQUIRE(xdw_train_supported(Cin, Cexp, stride, 1), "xdw_bwd_reduce: unsupported shape Cin=%d Cexp=%d s=%d", Cin, Cexp, stride);
    XtArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.w_exp = w_exp; a.Cexp = Cexp;
    a.sc_e = sc_e; a.sh_e = sh_e; a.mean_e = mean_e; a.rstd_e = rstd_e; a.act_e = act_e; a.w_dw = w_dw; a.dz_d = dz_d;
    xt_geometry(a, stride);
    const int KP = (Cin + 15) / 16 * 16;
    int blocks = xdw_train_blocks(B, H, W, Cexp);
    if (blocks > a.n_tiles) blocks = a.n_tiles;
    a.part = scratch; a.part_stride = xt_part_stride(KP, Cexp);
    *rows_out = blocks; *stride_out = a.part_stride;
    return stride == 1 ? launch_xt<XT_BWD, 1>(a, blocks, st) : launch_xt<XT_BWD, 2>(a, blocks, st);
}

// The first block of the network: stem conv (3x3 stride 2 on the normalised frame, 3 -> 32) as the "expand" layer over its 27-tap patch,
// followed by the first depthwise conv (stride 1).  There is no input gradient: this one pass yields everything the stem and the
// depthwise layer need.  Partial rows  S [2][32] | dWd [9][32] | G1 [32][32] | XX [32][32] | g0 [32]
size_t xdw_stem_scratch(int B, int fH, int fW) {
    int H, W, p;
    same_pad(fH + 1, 3, 2, 1, &H, &p);
    same_pad(fW + 1, 3, 2, 1, &W, &p);
    const int64_t stride = xt_part_stride(32, 32) + 32 * 32 + 32;
    return (size_t)(xdw_train_blocks(B, H, W, 32) + 1) * stride;
}

int launch_xdw_bwd_reduce_stem(const void* frames, int dtype, int B, int fH, int fW, float pixel_scale, const float* w_stem, const float* sc_e,
                               const float* sh_e, const float* mean_e, const float* rstd_e, int act_e, const float* w_dw, const float* dz_d,
                               float* scratch, int* rows_out, int64_t* stride_out, hipStream_t st) {
    AMS_REQUIRE(dtype == AMS_DT_U8 || dtype == AMS_DT_F32, "xdw_stem: frames must be uint8 or float32");
    XtArgs a;
    memset(&a, 0, sizeof(a));
    a.frames = frames; a.fH = fH; a.fW = fW; a.ps = pixel_scale; a.B = B; a.Cin = 27; a.w_exp = w_stem; a.Cexp = 32;
    same_pad(fH + 1, 3, 2, 1, &a.H, &a.spt);           // the stem's output grid = the depthwise conv's input grid
    same_pad(fW + 1, 3, 2, 1, &a.W, &a.spl);
    AMS_REQUIRE((int64_t)fH * fW * 3 < 0x7fffffffLL, "xdw_stem: frame too large");
    a.sc_e = sc_e; a.sh_e = sh_e; a.mean_e = mean_e; a.rstd_e = rstd_e; a.act_e = act_e; a.w_dw = w_dw; a.dz_d = dz_d;
    xt_geometry(a, 1);
    int blocks = xdw_train_blocks(B, a.H, a.W, 32);
    if (blocks > a.n_tiles) blocks = a.n_tiles;
    a.part = scratch; a.part_stride = xt_part_stride(32, 32) + 32 * 32 + 32;
    *rows_out = blocks; *stride_out = a.part_stride;
    const size_t lds = ((size_t)2 * (XtGeo<1>::TAP_FLOATS + 3 * 16 * 20) + 13 * 32) * sizeof(float);
    note_kernel("xdw_train_kernel<BWD, stem>");
    if (dtype == AMS_DT_U8) hipLaunchKernelGGL((xdw_train_kernel<XT_BWD, 1, 2, uint8_t>), dim3(blocks), dim3(128), lds, st, a);
    else hipLaunchKernelGGL((xdw_train_kernel<XT_BWD, 1, 2, float>), dim3(blocks), dim3(128), lds, st, a);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// backward pass 2: dx [B,H,W,Cin] = dz_e . W_e^T (+ res)
int launch_xdw_bwd_dx(const float* x, int B, int H, int W, int Cin, const float* w_exp, int Cexp, const float* sc_e, const float* sh_e,
                      int act_e, const float* w_dw, int stride, const float* dz_d, const float* cA, const float* cB, const float* cC,
                      const float* res, float* dx, hipStream_t st, const float* red_z, const float* red_mean, const float* red_rstd,
                      float* red_part, int* red_rows_out) {
    AMS_REQUIRE(xdw_train_supported(Cin, Cexp, stride, 1), "xdw_bwd_dx: unsupported shape Cin=%d Cexp=%d s=%d", Cin, Cexp, stride);
    AMS_REQUIRE((int64_t)H * W * (Cexp > Cin ? Cexp : Cin) < 0x7fffffffLL, "xdw_bwd_dx: image too large for 32-bit element offsets");
    XtArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.w_exp = w_exp; a.Cexp = Cexp;
    a.sc_e = sc_e; a.sh_e = sh_e; a.act_e = act_e; a.w_dw = w_dw; a.dz_d = dz_d; a.cA = cA; a.cB = cB; a.cC = cC; a.res = res; a.dx = dx;
    xt_geometry(a, stride);
    int64_t blocks = cdiv64(a.n_tiles, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (red_rows_out) *red_rows_out = 0;
    if (red_part) {
        AMS_REQUIRE(red_z && red_mean && red_rstd && red_rows_out && Cin % 4 == 0, "xdw_bwd_dx: incomplete arguments for the fused BN-backward sums");
        a.red_z = red_z; a.red_mean = red_mean; a.red_rstd = red_rstd; a.red_part = red_part;
        *red_rows_out = (int)blocks;
    }
    const size_t lds = ((size_t)14 * Cexp + 4 * (stride == 1 ? XtGeo<1>::TAP_FLOATS : XtGeo<2>::TAP_FLOATS)) * sizeof(float);
    const int KC = (Cin + 15) / 16;
    note_kernel("xdw_dx_kernel");
    if (stride == 1) {
        if (KC == 1) hipLaunchKernelGGL((xdw_dx_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((xdw_dx_kernel<1, 2>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    } else {
        if (KC == 1) hipLaunchKernelGGL((xdw_dx_kernel<2, 1>), dim3((unsigned)blocks), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((xdw_dx_kernel<2, 2>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    }
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_xdw_dwe(const float* G1, const float* xx_g0, int Cin, int Cexp, const float* w_exp, const float* cA, const float* cB, const float* cC,
                   float* dw, hipStream_t st) {
    const int KP = (Cin + 15) / 16 * 16;
    hipLaunchKernelGGL(xdw_dwe_kernel, dim3(cdiv(Cin * Cexp, 256)), dim3(256), 0, st, G1, xx_g0, KP, Cin, Cexp, w_exp, cA, cB, cC, dw);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

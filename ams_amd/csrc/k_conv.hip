// Stem (dense 3x3 stride 2 on the raw frame) and depthwise 3x3 kernels of the AMS student, NHWC, gfx950.
//
// These layers have 3.6-9 FLOP per activation element moved: pure HBM streaming.  Lanes run along the channel
// axis first (float4 = 4 channels per lane), then along x, so a wave's 64 lanes read/write 1 KiB of contiguous
// NHWC memory per instruction; the 3x3 halo re-reads are served by L1/L2, and xcd_remap() keeps spatially
// adjacent blocks on one XCD so the halo lines are fetched from HBM once.
#include "kernels.hpp"

namespace ams {

// ---------------------------------------------------------------------------------------------------------
// K1+K2  frame -> pad one row/col of 127.5 (nodes concat, concat_1) -> x*(1/127.5) - 1 (mul_4, sub_2)
//        -> Conv2D 3x3 stride 2 SAME (MobilenetV2/Conv) -> scale/shift (folded BN) -> act
// thread = (output pixel, group of 8 output channels)
// ---------------------------------------------------------------------------------------------------------
template <typename TIn>
__device__ __forceinline__ float load_px(const TIn* p) { return (float)(*p); }

template <typename TIn>
__device__ __forceinline__ float norm_frame_value(const TIn* img, int H, int W, int iy, int ix, int ch, float ps) {
    // coordinates are in the 127.5-padded (H+1)x(W+1) image; outside of it: SAME zero padding.
    // Branch-free (clamped address + selects): a guarded load would be an exec-masked VMEM op that hipcc waits for
    // with vmcnt(0), serialising the taps.
    const bool inside = iy >= 0 && ix >= 0 && iy <= H && ix <= W;
    const bool pad = iy >= H || ix >= W;
    const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const int ixc = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    float raw = load_px(img + ((int64_t)iyc * W + ixc) * 3 + ch);
    raw = pad ? 127.5f : raw;
    const float v = __fsub_rn(__fmul_rn(raw, ps), 1.0f);     // two roundings like the graph's Mul then Sub (no FMA contraction)
    return inside ? v : 0.f;
}

template <typename TIn>
__global__ __launch_bounds__(256) void stem_conv_kernel(const TIn* __restrict__ frames, int B, int H, int W,
                                                        const float* __restrict__ wgt, int cout,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int act, float ps, float* __restrict__ y, int Ho, int Wo, int pt,
                                                        int pl) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // [27][cout]
    for (int e = threadIdx.x; e < 27 * cout; e += blockDim.x) sw[e] = wgt[e];
    __syncthreads();
    const int groups = cout >> 3;
    const int64_t total = (int64_t)B * Ho * Wo * groups;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        int64_t p = t / groups;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const TIn* img = frames + (int64_t)b * H * W * 3;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = oy * 2 - pt + i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ix = ox * 2 - pl + j;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float v = norm_frame_value(img, H, W, iy, ix, ch, ps);
                    const float* wr = sw + ((i * 3 + j) * 3 + ch) * cout + g * 8;
                    const float4 w0 = ld4(wr), w1 = ld4(wr + 4);
                    acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]);
                    acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                    acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]);
                    acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
                }
            }
        }
        float* out = y + t * 8;
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = acc[c];
            if (scale) v = v * scale[g * 8 + c] + shift[g * 8 + c];
            o[c] = apply_act(v, act);
        }
        st4(out, make_float4(o[0], o[1], o[2], o[3]));
        st4(out + 4, make_float4(o[4], o[5], o[6], o[7]));
    }
}

// MFMA form of the same layer (cout = 32): the 3x3x3 receptive field is a K = 27 (padded to 32) contraction, i.e. a
// [pixels, 32] x [32, 32] GEMM whose A operand is gathered straight from the uint8 frame.  Exact f32 MFMA with the
// operand roles swapped as in k_pointwise.hip: lane (p = lane & 15, q = lane >> 4) supplies taps k = 16c + 4q + j of
// pixel p and ends up owning output channels 16t + 4q .. +3 of that pixel (float4 epilogue).  8 byte-gathers and
// 16 MFMAs per 16 pixels instead of 27 gathers and 864 scalar FMAs per pixel.
template <typename TIn>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const TIn* __restrict__ frames, int B, int H, int W,
                                                        const float* __restrict__ wgt, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int act, float ps,
                                                        float* __restrict__ y, int Ho, int Wo, int pt, int pl, int64_t n_groups) {
    constexpr int PITCH = 36, RM = 2, NT = 2;
    __shared__ float sW[32 * PITCH];
    __shared__ __attribute__((aligned(16))) float sOut[4 * 16 * 36];
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
        const int kk = e >> 5, nn = e & 31;
        sW[kk * PITCH + nn] = kk < 27 ? wgt[kk * 32 + nn] : 0.f;
    }
    __syncthreads();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    // this lane's 8 taps: k = 16c + 4q + j  ->  (dy, dx, ch); k >= 27 are padding
    int tdy[8], tdx[8], tch[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = 16 * (u >> 2) + 4 * q + (u & 3);
        const int tap = k / 3;
        tch[u] = k < 27 ? k - tap * 3 : -1;
        tdy[u] = tap / 3;
        tdx[u] = tap - tdy[u] * 3;
    }
    const int64_t total_px = (int64_t)B * Ho * Wo;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < n_groups; g += (int64_t)gridDim.x * 4) {
        f32x4 acc[RM][NT];
        float v[RM][8];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            int p = (int)(g * (16 * RM)) + r * 16 + l15;                // total_px < 2^31 (checked on the host): 32-bit decode
            if (p > (int)total_px - 1) p = (int)total_px - 1;
            const int ox = p % Wo;
            const int t2 = p / Wo;
            const int oy = t2 % Ho, b = t2 / Ho;
            const TIn* img = frames + (int64_t)b * H * W * 3;
            const int iy0 = oy * 2 - pt, ix0 = ox * 2 - pl;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                {
                    const float t = norm_frame_value(img, H, W, iy0 + tdy[u], ix0 + tdx[u], tch[u] < 0 ? 0 : tch[u], ps);
                    v[r][u] = tch[u] >= 0 ? t : 0.f;
                }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* sB = sW + (16 * (u >> 2) + 4 * q + (u & 3)) * PITCH + l15;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float wv = sB[16 * t];
#pragma unroll
                for (int r = 0; r < RM; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, v[r][u], acc[r][t], 0, 0, 0);
            }
        }
        // epilogue: BN + activation, then through a per-wave LDS slab so that the 16 pixels x 128 B of a row group leave
        // as 2 KB of consecutive float4 (a lane's MFMA result is 4 channels of ONE pixel: direct stores would hit 16
        // different lines in 64-byte pieces)
        float* slab = sOut + wave * (16 * 36);
#pragma unroll
        for (int r = 0; r < RM; ++r) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n4 = 16 * t + 4 * q;
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (scale) { sc = ld4(scale + n4); sh = ld4(shift + n4); }
                float4 o;
                o.x = apply_act(acc[r][t][0] * sc.x + sh.x, act); o.y = apply_act(acc[r][t][1] * sc.y + sh.y, act);
                o.z = apply_act(acc[r][t][2] * sc.z + sh.z, act); o.w = apply_act(acc[r][t][3] * sc.w + sh.w, act);
                st4(slab + l15 * 36 + n4, o);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t p0 = g * (16 * RM) + r * 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = lane + 64 * u, row = f >> 3, c4 = (f & 7) * 4;
                const float4 o = ld4(slab + row * 36 + c4);
                if (p0 + row < total_px) st4(y + (p0 + row) * 32 + c4, o);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

int launch_stem(const void* frames, int dtype, int B, int H, int W, const float* w, int cout, const float* scale,
                const float* shift, int act, float pixel_scale, float* y, hipStream_t st) {
    AMS_REQUIRE(cout % 8 == 0 && cout <= 256, "stem: cout %d must be a multiple of 8", cout);
    AMS_REQUIRE(dtype == AMS_DT_U8 || dtype == AMS_DT_F32, "stem: frames must be uint8 or float32");
    AMS_REQUIRE((scale == nullptr) == (shift == nullptr), "stem: scale and shift come together");
    int Ho, Wo, pt, pl;
    same_pad(H + 1, 3, 2, 1, &Ho, &pt);
    same_pad(W + 1, 3, 2, 1, &Wo, &pl);
    if (cout == 32) {
        AMS_REQUIRE((int64_t)B * Ho * Wo < 0x7fffffffLL, "stem: too many pixels");
        const int64_t n_groups = cdiv64((int64_t)B * Ho * Wo, 32);
        int64_t grid = cdiv64(n_groups, 4);
        if (grid > 256 * 8) grid = 256 * 8;
        note_kernel(dtype == AMS_DT_U8 ? "stem_mfma_kernel<unsigned char>" : "stem_mfma_kernel<float>");
        if (dtype == AMS_DT_U8)
            hipLaunchKernelGGL(stem_mfma_kernel<uint8_t>, dim3((unsigned)grid), dim3(256), 0, st, (const uint8_t*)frames, B, H, W, w,
                               scale, shift, act, pixel_scale, y, Ho, Wo, pt, pl, n_groups);
        else
            hipLaunchKernelGGL(stem_mfma_kernel<float>, dim3((unsigned)grid), dim3(256), 0, st, (const float*)frames, B, H, W, w,
                               scale, shift, act, pixel_scale, y, Ho, Wo, pt, pl, n_groups);
        AMS_CHECK_LAUNCH();
        return AMS_OK;
    }
    const int64_t total = (int64_t)B * Ho * Wo * (cout / 8);
    const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
    const size_t lds = 27 * cout * sizeof(float);
    note_kernel(dtype == AMS_DT_U8 ? "stem_conv_kernel<unsigned char>" : "stem_conv_kernel<float>");
    if (dtype == AMS_DT_U8)
        hipLaunchKernelGGL(stem_conv_kernel<uint8_t>, dim3(grid), dim3(256), lds, st, (const uint8_t*)frames, B, H, W, w, cout,
                           scale, shift, act, pixel_scale, y, Ho, Wo, pt, pl);
    else
        hipLaunchKernelGGL(stem_conv_kernel<float>, dim3(grid), dim3(256), lds, st, (const float*)frames, B, H, W, w, cout,
                           scale, shift, act, pixel_scale, y, Ho, Wo, pt, pl);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// im2col of the stem's receptive fields, [B*Ho*Wo, 32] (27 taps in (i,j,ch) order + 5 zero columns): the stem's
// weight gradient then is the generic x^T @ dy GEMM.  thread = (pixel, 4 columns)
template <typename TIn>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const TIn* __restrict__ frames, int B, int H, int W, float ps,
                                                          float* __restrict__ out, int Ho, int Wo, int pt, int pl) {
    const int64_t total = (int64_t)B * Ho * Wo * 8;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t & 7);
        int64_t p = t >> 3;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const TIn* img = frames + (int64_t)b * H * W * 3;
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = g * 4 + c;
            float v = 0.f;
            if (col < 27) {
                const int tap = col / 3, ch = col % 3;
                v = norm_frame_value(img, H, W, oy * 2 - pt + tap / 3, ox * 2 - pl + tap % 3, ch, ps);
            }
            o[c] = v;
        }
        st4(out + t * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

int launch_stem_im2col(const void* frames, int dtype, int B, int H, int W, float pixel_scale, float* out, hipStream_t st) {
    AMS_REQUIRE(dtype == AMS_DT_U8 || dtype == AMS_DT_F32, "stem_im2col: frames must be uint8 or float32");
    int Ho, Wo, pt, pl;
    same_pad(H + 1, 3, 2, 1, &Ho, &pt);
    same_pad(W + 1, 3, 2, 1, &Wo, &pl);
    const int64_t total = (int64_t)B * Ho * Wo * 8;
    const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
    note_kernel("stem_im2col_kernel");
    if (dtype == AMS_DT_U8)
        hipLaunchKernelGGL(stem_im2col_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)frames, B, H, W,
                           pixel_scale, out, Ho, Wo, pt, pl);
    else
        hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)frames, B, H, W,
                           pixel_scale, out, Ho, Wo, pt, pl);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K3  depthwise 3x3, stride S in {1,2}, rate R in {1,2} (rate 2 == the graph's SpaceToBatchND / VALID /
//     BatchToSpaceND sandwich), SAME padding, fused scale/shift + activation.
// block = CG x SLOTS threads: CG = C/4 channel groups (lanes first), SLOTS adjacent output columns;
// each thread produces TH consecutive output rows for its (column, channel group).
// ---------------------------------------------------------------------------------------------------------
struct DwGeom {
    int B, H, W, C, Ho, Wo, pt, pl, CG, slots, TH, tiles_x, tiles_y;
    int flip;      // forward kernel: taps taken in reverse order (the input gradient of a stride-1 conv is the conv with the flipped kernel)
};

static int dw_geom(int B, int H, int W, int C, int stride, int rate, bool over_input, DwGeom* g) {
    AMS_REQUIRE(C % 4 == 0 && C / 4 <= 256, "depthwise: C=%d must be a multiple of 4 and <= 1024", C);
    AMS_REQUIRE((stride == 1 || stride == 2) && (rate == 1 || rate == 2) && !(stride == 2 && rate == 2),
                "depthwise: unsupported stride %d / rate %d", stride, rate);
    g->B = B; g->H = H; g->W = W; g->C = C; g->flip = 0;
    same_pad(H, 3, stride, rate, &g->Ho, &g->pt);
    same_pad(W, 3, stride, rate, &g->Wo, &g->pl);
    g->CG = C / 4;
    g->slots = 256 / g->CG < 1 ? 1 : 256 / g->CG;
    g->TH = 4;
    const int rows = over_input ? H : g->Ho, cols = over_input ? W : g->Wo;
    g->tiles_x = cdiv(cols, g->slots);
    g->tiles_y = cdiv(rows, g->TH);
    return AMS_OK;
}

// Forward: a thread owns one output column (4 channels) and TH consecutive output rows.  All distinct input taps of
// the TH rows ((TH-1)*S + 2*R + 1 input rows x 3 columns) are requested up front — independent loads, nothing
// serialised — and each is used by every output row it belongs to (4.5 loads per output for stride 1 instead of 9).
template <int S, int R, int TH>
__global__ __launch_bounds__(256) void dw3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int act, float* __restrict__ y, DwGeom g, unsigned nblocks) {
    // Rate 2 at stride 1 is two interleaved ordinary convolutions on the even and the odd rows: a thread takes TH output
    // rows of ONE parity (row step RS = 2), which touch TH + 2 input rows of that parity instead of TH + 4 consecutive rows
    // (18 loads per 4 outputs instead of 24).  tiles_y then counts (row block, parity) pairs.
    constexpr int RS = (S == 1 && R == 2) ? 2 : 1;          // step between a thread's output rows
    constexpr int US = RS == 2 ? R : 1;                     // step between the input rows it loads
    constexpr int NR = RS == 2 ? TH + 2 : (TH - 1) * S + 2 * R + 1;        // input rows touched by the TH output rows
    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int tx = lb % g.tiles_x;
    const int ty = (lb / g.tiles_x) % g.tiles_y;
    const int b = lb / (g.tiles_x * g.tiles_y);
    // flat (column, channel group) index across the row: every wave is full whatever C/4 is (a block of CG x slots threads
    // left the third wave of the 576-channel layers three quarters empty)
    const int flat = tx * 256 + threadIdx.x;
    const int ox = flat / g.CG, cg = flat - ox * g.CG;
    if (ox >= g.Wo) return;
    const int c0 = cg * 4;
    const float* xb = x + (int64_t)b * g.H * g.W * g.C + c0;
    float* yb = y + (int64_t)b * g.Ho * g.Wo * g.C + c0;
    const int oy0 = (ty / RS) * (TH * RS) + (ty % RS);
    const int iy0 = oy0 * S - g.pt;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int ixc[3];
    bool okx[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int ix = ox * S - g.pl + j * R;
        okx[j] = ix >= 0 && ix < g.W;
        ixc[j] = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
    }
    float4 in[NR][3];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int iy = iy0 + rr * US;
        const bool oky = iy >= 0 && iy < g.H;
        const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
        const float* rp = xb + (int64_t)iyc * g.W * g.C;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // clamped address + select instead of a guarded load: no exec-masked VMEM, all taps stay in flight together
            const float4 v = ld4(rp + (int64_t)ixc[j] * g.C);
            const bool ok = oky && okx[j];
            in[rr][j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    }
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(wgt + (g.flip ? 8 - k : k) * g.C + c0);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero4;
    if (scale) { sc = ld4(scale + c0); sh = ld4(shift + c0); }
#pragma unroll
    for (int r = 0; r < TH; ++r) {
        const int oy = oy0 + r * RS;
        if (oy >= g.Ho) break;
        float4 acc = zero4;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 v = in[RS == 2 ? r + i : r * S + i * R][j];
                const float4 w4 = wv[i * 3 + j];
                acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
            }
        float4 o;
        o.x = apply_act(acc.x * sc.x + sh.x, act); o.y = apply_act(acc.y * sc.y + sh.y, act);
        o.z = apply_act(acc.z * sc.z + sh.z, act); o.w = apply_act(acc.w * sc.w + sh.w, act);
        st4(yb + ((int64_t)oy * g.Wo + ox) * g.C, o);
    }
}

int launch_depthwise(const float* x, int B, int H, int W, int C, const float* w, int stride, int rate,
                     const float* scale, const float* shift, int act, float* y, hipStream_t st) {
    DwGeom g;
    int rc = dw_geom(B, H, W, C, stride, rate, false, &g);
    if (rc) return rc;
    AMS_REQUIRE((scale == nullptr) == (shift == nullptr), "depthwise: scale and shift come together");
    constexpr int TH1 = 4, TH2 = 3, THR = 4;
    g.tiles_y = stride == 2 ? cdiv(g.Ho, TH2) : rate == 2 ? 2 * cdiv(g.Ho, 2 * THR) : cdiv(g.Ho, TH1);     // rate 2: (row block, parity)
    g.tiles_x = cdiv(g.Wo * g.CG, 256);
    const unsigned nblocks = (unsigned)g.tiles_x * g.tiles_y * B;
    const int threads = 256;
    note_kernel(stride == 2 ? "dw3x3_fwd_kernel<2, 1, 3>" : rate == 2 ? "dw3x3_fwd_kernel<1, 2, 4>" : "dw3x3_fwd_kernel<1, 1, 4>");
    if (stride == 1 && rate == 1)
        hipLaunchKernelGGL((dw3x3_fwd_kernel<1, 1, TH1>), dim3(nblocks), dim3(threads), 0, st, x, w, scale, shift, act, y, g, nblocks);
    else if (stride == 2)
        hipLaunchKernelGGL((dw3x3_fwd_kernel<2, 1, TH2>), dim3(nblocks), dim3(threads), 0, st, x, w, scale, shift, act, y, g, nblocks);
    else
        hipLaunchKernelGGL((dw3x3_fwd_kernel<1, 2, THR>), dim3(nblocks), dim3(threads), 0, st, x, w, scale, shift, act, y, g, nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// input gradient: dx[iy,ix,c] = sum_{i,j} dy[oy,ox,c] * w[i,j,c]  with  oy*S - pt + i*R == iy  (same for x)
template <int S, int R>
__global__ __launch_bounds__(256) void dw3x3_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wgt,
                                                          float* __restrict__ dx, DwGeom g, unsigned nblocks) {
    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int tx = lb % g.tiles_x;
    const int ty = (lb / g.tiles_x) % g.tiles_y;
    const int b = lb / (g.tiles_x * g.tiles_y);
    const int flat = tx * 256 + threadIdx.x;               // flat (column, channel group) index: full waves for any C/4
    const int ix = flat / g.CG, cg = flat - ix * g.CG;
    if (ix >= g.W) return;
    const int c0 = cg * 4;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(wgt + k * g.C + c0);
    const float* dyb = dy + (int64_t)b * g.Ho * g.Wo * g.C + c0;
    float* dxb = dx + (int64_t)b * g.H * g.W * g.C + c0;
#pragma unroll 1
    for (int r = 0; r < g.TH; ++r) {
        const int iy = ty * g.TH + r;
        if (iy >= g.H) break;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ny = iy + g.pt - i * R;
            const bool oky = ny >= 0 && !(S == 2 && (ny & 1)) && ny / S < g.Ho;
            int oy = ny < 0 ? 0 : ny / S;
            if (oy > g.Ho - 1) oy = g.Ho - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int nx = ix + g.pl - j * R;
                const bool ok = oky && nx >= 0 && !(S == 2 && (nx & 1)) && nx / S < g.Wo;
                int ox = nx < 0 ? 0 : nx / S;
                if (ox > g.Wo - 1) ox = g.Wo - 1;
                const float4 v = ld4(dyb + ((int64_t)oy * g.Wo + ox) * g.C);      // clamped address, masked by select
                const float4 w4 = wv[i * 3 + j];
                acc.x = fmaf(ok ? v.x : 0.f, w4.x, acc.x); acc.y = fmaf(ok ? v.y : 0.f, w4.y, acc.y);
                acc.z = fmaf(ok ? v.z : 0.f, w4.z, acc.z); acc.w = fmaf(ok ? v.w : 0.f, w4.w, acc.w);
            }
        }
        st4(dxb + ((int64_t)iy * g.W + ix) * g.C, acc);
    }
}

// Stride 2: an input pixel receives 1, 2 or 4 of the 9 taps, depending on the parity of its row and column.  A thread takes the
// 2x2 block of input pixels whose top-left pixel has (row + pad) and (column + pad) even: the block needs exactly 2x2 gradient
// values and 9 products in all (the generic kernel loads 36 values for the same four pixels, 27 of them masked); per pixel the
// products come in the generic kernel's order (i, then j), so the result is bit-identical.
__global__ __launch_bounds__(256) void dw3x3_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ wgt,
                                                             float* __restrict__ dx, DwGeom g, int blocks_x, int blocks_y, int th) {
    const int tiles_x = (blocks_x * g.CG + 255) / 256;
    const int tx = blockIdx.x % tiles_x;
    const int ty = (blockIdx.x / tiles_x) % ((blocks_y + th - 1) / th);
    const int b = blockIdx.x / (tiles_x * ((blocks_y + th - 1) / th));
    const int flat = tx * 256 + threadIdx.x;
    const int bx = flat / g.CG, cg = flat - bx * g.CG;
    if (bx >= blocks_x) return;
    const int c0 = cg * 4;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(wgt + k * g.C + c0);
    const float* dyb = dy + (int64_t)b * g.Ho * g.Wo * g.C + c0;
    float* dxb = dx + (int64_t)b * g.H * g.W * g.C + c0;
    const int ix0 = 2 * bx - g.pl;                          // ix0 + pl even
    const int oxH = (ix0 + g.pl) / 2, oxL = oxH - 1;
    const bool okxH = oxH < g.Wo, okxL = oxL >= 0;
    const int oxHc = okxH ? oxH : g.Wo - 1, oxLc = okxL ? oxL : 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int r = 0; r < th; ++r) {
        const int by = ty * th + r;
        if (by >= blocks_y) break;
        const int iy0 = 2 * by - g.pt;
        const int oyH = (iy0 + g.pt) / 2, oyL = oyH - 1;
        const bool okyH = oyH < g.Ho, okyL = oyL >= 0;
        const int oyHc = okyH ? oyH : g.Ho - 1, oyLc = okyL ? oyL : 0;
        float4 dHH = ld4(dyb + ((int64_t)oyHc * g.Wo + oxHc) * g.C), dHL = ld4(dyb + ((int64_t)oyHc * g.Wo + oxLc) * g.C);
        float4 dLH = ld4(dyb + ((int64_t)oyLc * g.Wo + oxHc) * g.C), dLL = ld4(dyb + ((int64_t)oyLc * g.Wo + oxLc) * g.C);
        if (!(okyH && okxH)) dHH = z4;
        if (!(okyH && okxL)) dHL = z4;
        if (!(okyL && okxH)) dLH = z4;
        if (!(okyL && okxL)) dLL = z4;
        auto fma4 = [](float4 a, const float4& v, const float4& w4) {
            a.x = fmaf(v.x, w4.x, a.x); a.y = fmaf(v.y, w4.y, a.y); a.z = fmaf(v.z, w4.z, a.z); a.w = fmaf(v.w, w4.w, a.w);
            return a;
        };
        // pixel (A, A): taps (0,0) (0,2) (2,0) (2,2); (A, B): (0,1) (2,1); (B, A): (1,0) (1,2); (B, B): (1,1)
        float4 aAA = fma4(fma4(fma4(fma4(z4, dHH, wv[0]), dHL, wv[2]), dLH, wv[6]), dLL, wv[8]);
        float4 aAB = fma4(fma4(z4, dHH, wv[1]), dLH, wv[7]);
        float4 aBA = fma4(fma4(z4, dHH, wv[3]), dHL, wv[5]);
        float4 aBB = fma4(z4, dHH, wv[4]);
        const bool rA = iy0 >= 0 && iy0 < g.H, rB = iy0 + 1 < g.H, cA = ix0 >= 0 && ix0 < g.W, cB = ix0 + 1 < g.W;
        if (rA && cA) st4(dxb + ((int64_t)iy0 * g.W + ix0) * g.C, aAA);
        if (rA && cB) st4(dxb + ((int64_t)iy0 * g.W + ix0 + 1) * g.C, aAB);
        if (rB && cA) st4(dxb + ((int64_t)(iy0 + 1) * g.W + ix0) * g.C, aBA);
        if (rB && cB) st4(dxb + ((int64_t)(iy0 + 1) * g.W + ix0 + 1) * g.C, aBB);
    }
}

int launch_depthwise_dgrad(const float* dy, int B, int H, int W, int C, const float* w, int stride, int rate,
                           float* dx, hipStream_t st) {
    DwGeom g;
    int rc = dw_geom(B, H, W, C, stride, rate, true, &g);
    if (rc) return rc;
    if (stride == 1) {
        // dx[iy] = sum_i dy[iy + R - i R] w[i] = sum_i' dy[iy - R + i' R] w[2 - i'] (SAME padding is symmetric at stride 1): the
        // forward kernel with the taps reversed — its row reuse (4.5 loads per output instead of 9) and its HBM rate
        DwGeom f;
        rc = dw_geom(B, H, W, C, 1, rate, false, &f);
        if (rc) return rc;
        f.flip = 1;
        constexpr int TH1 = 4, THR = 4;
        f.tiles_y = rate == 2 ? 2 * cdiv(f.Ho, 2 * THR) : cdiv(f.Ho, TH1);
        f.tiles_x = cdiv(f.Wo * f.CG, 256);
        const unsigned nb = (unsigned)f.tiles_x * f.tiles_y * B;
        note_kernel(rate == 2 ? "dw3x3_fwd_kernel<1, 2, 4>" : "dw3x3_fwd_kernel<1, 1, 4>");
        if (rate == 1)
            hipLaunchKernelGGL((dw3x3_fwd_kernel<1, 1, TH1>), dim3(nb), dim3(256), 0, st, dy, w, (const float*)nullptr, (const float*)nullptr,
                               (int)AMS_ACT_NONE, dx, f, nb);
        else
            hipLaunchKernelGGL((dw3x3_fwd_kernel<1, 2, THR>), dim3(nb), dim3(256), 0, st, dy, w, (const float*)nullptr, (const float*)nullptr,
                               (int)AMS_ACT_NONE, dx, f, nb);
        AMS_CHECK_LAUNCH();
        return AMS_OK;
    }
    if (stride == 2) {
        // 2x2 input blocks aligned to the padding: block (by, bx) starts at (2 by - pt, 2 bx - pl)
        const int blocks_y = (H + g.pt + 1) / 2, blocks_x = (W + g.pl + 1) / 2, th = 4;
        const unsigned nb = (unsigned)cdiv((int64_t)blocks_x * g.CG, 256) * cdiv(blocks_y, th) * B;
        note_kernel("dw3x3_dgrad_s2_kernel");
        hipLaunchKernelGGL(dw3x3_dgrad_s2_kernel, dim3(nb), dim3(256), 0, st, dy, w, dx, g, blocks_x, blocks_y, th);
        AMS_CHECK_LAUNCH();
        return AMS_OK;
    }
    g.tiles_x = cdiv(g.W * g.CG, 256);
    const unsigned nblocks = (unsigned)g.tiles_x * g.tiles_y * B;
    const int threads = 256;
    note_kernel(stride == 2 ? "dw3x3_dgrad_kernel<2, 1>" : rate == 2 ? "dw3x3_dgrad_kernel<1, 2>" : "dw3x3_dgrad_kernel<1, 1>");
    if (stride == 1 && rate == 1)
        hipLaunchKernelGGL((dw3x3_dgrad_kernel<1, 1>), dim3(nblocks), dim3(threads), 0, st, dy, w, dx, g, nblocks);
    else if (stride == 2)
        hipLaunchKernelGGL((dw3x3_dgrad_kernel<2, 1>), dim3(nblocks), dim3(threads), 0, st, dy, w, dx, g, nblocks);
    else
        hipLaunchKernelGGL((dw3x3_dgrad_kernel<1, 2>), dim3(nblocks), dim3(threads), 0, st, dy, w, dx, g, nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// Input gradient of a stride-1 depthwise conv FUSED with the first half of the BN backward of the layer in front of it (the fine-tune
// step of the stride-16 blocks: that layer is the block's expand layer).  da_e = dwconv^T(dz_d) as above; then, with the expand layer's
// raw output z_e read at the same position, dy_e = da_e . act'(z_e sc + sh) is what gets written, and sum(dy_e), sum(dy_e xhat_e) are
// accumulated on the way: the separate reduction pass over (da_e, z_e) — one launch and two tensor reads per block — disappears, and
// bn_bwd_apply then runs on dy_e with the activation already applied.  The depthwise WEIGHT gradient rides along too: tap (i, j) pairs
// a_e at this position (= clamp(z_e sc + sh), already at hand for the mask) with exactly the dz_d value that the input gradient multiplies
// by w[i][j], so dW[i][j] += a_e . dz_d costs nine more FMAs per output and no load — the separate weight-gradient kernel (two tensor
// reads) and its split reduction disappear from the side stream.  A block owns one column strip of one image and walks DOWN all its
// row tiles with its sums in registers; threads of a block that share a channel group are added in a fixed order through LDS:
// one partial row [2 + 9][C] per block, deterministic.
template <int R>
__global__ __launch_bounds__(256) void dw3x3_dgrad_bn_kernel(const float* __restrict__ dy, const float* __restrict__ wgt, const float* __restrict__ z,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ out, float* __restrict__ part, DwGeom g, int bands,
                                                             int tiles_per_band) {
    constexpr int TH = 4, RS = R == 2 ? 2 : 1, NR = RS == 2 ? TH + 2 : TH + 2 * R;
    constexpr int NQ = 8 + 36;                                         // s1, s2, nine taps: 4 channels each
    __shared__ float s_acc[256][NQ + 1];
    const int tx = blockIdx.x % g.tiles_x, band = (blockIdx.x / g.tiles_x) % bands, b = blockIdx.x / (g.tiles_x * bands);
    const int flat = tx * 256 + threadIdx.x;
    const int ox = flat / g.CG, cg = flat - ox * g.CG;
    const bool live = ox < g.Wo;
    const int c0 = cg * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = zero4, s2 = zero4, dwv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv[k] = zero4;
    if (live) {
        const float* xb = dy + (int64_t)b * g.H * g.W * g.C + c0;
        const float* zb = z + (int64_t)b * g.Ho * g.Wo * g.C + c0;
        float* yb = out + (int64_t)b * g.Ho * g.Wo * g.C + c0;
        int ixc[3];
        bool okx[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ix = ox - g.pl + j * R;
            okx[j] = ix >= 0 && ix < g.W;
            ixc[j] = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
        }
        float4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = ld4(wgt + (8 - k) * g.C + c0);            // the conv with the flipped kernel
        const float4 sc = ld4(scale + c0), sh = ld4(shift + c0), mu = ld4(mean + c0), rs = ld4(rstd + c0);
        const float lo = act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
        const int ty_end = (band + 1) * tiles_per_band < g.tiles_y ? (band + 1) * tiles_per_band : g.tiles_y;
        for (int ty = band * tiles_per_band; ty < ty_end; ++ty) {
            const int oy0 = (ty / RS) * (TH * RS) + (ty % RS);
            const int iy0 = oy0 - g.pt;
            float4 in[NR][3];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int iy = iy0 + rr * (RS == 2 ? R : 1);
                const bool oky = iy >= 0 && iy < g.H;
                const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
                const float* rp = xb + (int64_t)iyc * g.W * g.C;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 v = ld4(rp + (int64_t)ixc[j] * g.C);
                    const bool ok = oky && okx[j];
                    in[rr][j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
                }
            }
            float4 zv[TH];
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int oy = oy0 + r * RS;
                const int oyc = oy < g.Ho ? oy : g.Ho - 1;
                zv[r] = ld4(zb + ((int64_t)oyc * g.Wo + ox) * g.C);
            }
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int oy = oy0 + r * RS;
                if (oy >= g.Ho) break;
                float4 acc = zero4;
                const float4 y = muladd4_pk(zv[r], sc, sh);
                const float4 ae = make_float4(__builtin_amdgcn_fmed3f(y.x, lo, hi), __builtin_amdgcn_fmed3f(y.y, lo, hi),
                                              __builtin_amdgcn_fmed3f(y.z, lo, hi), __builtin_amdgcn_fmed3f(y.w, lo, hi));
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float4 v = in[RS == 2 ? r + i : r + i * R][j];
                        const float4 w4 = wv[i * 3 + j];
                        acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                        acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
                        // position (i, j) of the flipped kernel is tap 8 - (3i + j) of the forward conv
                        fma4_pk(dwv[8 - (i * 3 + j)], ae, v);
                    }
                const float4 d = make_float4((y.x > lo && y.x < hi) ? acc.x : 0.f, (y.y > lo && y.y < hi) ? acc.y : 0.f,
                                             (y.z > lo && y.z < hi) ? acc.z : 0.f, (y.w > lo && y.w < hi) ? acc.w : 0.f);
                st4(yb + ((int64_t)oy * g.Wo + ox) * g.C, d);
                s1 = add4_pk(s1, d);
                s2 = add4_pk(s2, mul4_pk(mul4_pk(d, sub4_pk(zv[r], mu)), rs));
            }
        }
    }
    // fixed-order sum of the threads that share a channel group (thread t, t + CG, ...), then one partial row per block
    float* sa = s_acc[threadIdx.x];
    sa[0] = s1.x; sa[1] = s1.y; sa[2] = s1.z; sa[3] = s1.w; sa[4] = s2.x; sa[5] = s2.y; sa[6] = s2.z; sa[7] = s2.w;
#pragma unroll
    for (int k = 0; k < 9; ++k) { sa[8 + 4 * k] = dwv[k].x; sa[9 + 4 * k] = dwv[k].y; sa[10 + 4 * k] = dwv[k].z; sa[11 + 4 * k] = dwv[k].w; }
    sa[NQ] = live ? 1.f : -1.f;
    __syncthreads();
    // a block touches at most min(CG, 256) channel groups: cg0 .. ; thread e owns (group index e / 8 of the block's range, quantity e % 8)
    const int cg_first = (tx * 256) % g.CG;
    const int ngroups = g.CG < 256 ? g.CG : 256;
    float* row = part + (int64_t)blockIdx.x * 11 * g.C;              // [sum dy | sum dy xhat | dW taps 0..8][C]
    for (int e = threadIdx.x; e < ngroups * NQ; e += 256) {
        const int gi = e / NQ, qn = e - gi * NQ;
        const int cgw = (cg_first + gi) % g.CG;
        float s = 0.f;
        for (int t = gi; t < 256; t += g.CG)                        // threads t with (tx * 256 + t) % CG == cgw, ascending
            if (s_acc[t][NQ] >= 0.f) s += s_acc[t][qn];
        row[(qn >> 2) * g.C + cgw * 4 + (qn & 3)] = s;
    }
}

// A strip per image is too few blocks on the narrow layers: the walk down an image is split into bands of row tiles until the launch has
// ~`target` blocks (one partial row per block either way).
static void dw_walk_plan(int B, int H, int W, int C, int rate, int target, DwGeom* f, int* bands, int* tpb) {
    dw_geom(B, H, W, C, 1, rate, false, f);
    f->tiles_y = rate == 2 ? 2 * cdiv(f->Ho, 8) : cdiv(f->Ho, 4);
    f->tiles_x = cdiv((int64_t)f->Wo * f->CG, 256);
    int nb = cdiv(target, f->tiles_x * B);
    nb = nb < 1 ? 1 : (nb > f->tiles_y ? f->tiles_y : nb);
    *tpb = cdiv(f->tiles_y, nb);
    *bands = cdiv(f->tiles_y, *tpb);
}
// backward kernel: bands measured no faster at 8 frames of 512x1024 (400 blocks: 9.13 ms a step, 768: 9.21, one strip per image: 9.11 —
// every extra partial row is eleven vectors for the second stage and the tap reduction), so it keeps one strip per image
constexpr int kDgradBnBlocks = 1;

size_t depthwise_dgrad_bn_scratch(int B, int H, int W, int C) {
    if (C % 4 != 0 || C / 4 > 256) return (size_t)-1;
    DwGeom f; int bands, tpb;
    // (the rate does not change the count: tiles_y only caps the bands, and both rates have >= 8 row tiles wherever bands are wanted)
    dw_walk_plan(B, H, W, C, 1, kDgradBnBlocks, &f, &bands, &tpb);
    DwGeom f2; int bands2, tpb2;
    dw_walk_plan(B, H, W, C, 2, kDgradBnBlocks, &f2, &bands2, &tpb2);
    return (size_t)B * f.tiles_x * (bands > bands2 ? bands : bands2) * 11 * C;
}

// dx_masked [B,H,W,C] = dwconv^T(dy, w) . act'(z sc + sh) (stride 1, rate 1|2; the conv keeps the size), partial rows [rows][11][C] of
// (sum dx_masked, sum dx_masked xhat, the nine taps of the depthwise weight gradient sum act(z sc + sh) . dy) in scratch
int launch_depthwise_dgrad_bn(const float* dy, int B, int H, int W, int C, const float* w, int rate, const float* z, const float* scale,
                              const float* shift, int act, const float* mean, const float* rstd, float* out, float* scratch, int* rows_out,
                              hipStream_t st) {
    DwGeom f;
    int rc = dw_geom(B, H, W, C, 1, rate, false, &f);
    if (rc) return rc;
    AMS_REQUIRE(C / 4 >= 1, "depthwise_dgrad_bn: C=%d", C);
    int bands, tpb;
    dw_walk_plan(B, H, W, C, rate, kDgradBnBlocks, &f, &bands, &tpb);
    const unsigned nb = (unsigned)f.tiles_x * bands * B;
    *rows_out = (int)nb;
    // a block whose 256 flat indices wrap around the channel groups more than once covers every group: rows are complete; otherwise the
    // groups it does not touch must read as zero
    AMS_REQUIRE(f.CG <= 256, "depthwise_dgrad_bn: C=%d exceeds 1024", C);
    note_kernel(rate == 2 ? "dw3x3_dgrad_bn_kernel<2>" : "dw3x3_dgrad_bn_kernel<1>");
    if (rate == 1) hipLaunchKernelGGL((dw3x3_dgrad_bn_kernel<1>), dim3(nb), dim3(256), 0, st, dy, w, z, scale, shift, act, mean, rstd, out, scratch, f, bands, tpb);
    else hipLaunchKernelGGL((dw3x3_dgrad_bn_kernel<2>), dim3(nb), dim3(256), 0, st, dy, w, z, scale, shift, act, mean, rstd, out, scratch, f, bands, tpb);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// Training forward of a stride-1 depthwise layer in a block that keeps its tensors, three passes in one: the taps are read from the
// expand layer's RAW output z_e with its BN + activation applied on the load (a_e = act(z_e sc + sh) is never written: the backward
// kernel above recomputes it the same way), the depthwise result z_d is written, and the BN statistics of z_d — shifted sums about
// `center` — are accumulated on the way.  Same walk as the backward kernel: a block owns one column strip of one image and a band of
// its row tiles, sums in registers, threads that share a channel group folded in a fixed order: one partial row [2][C] per block.
template <int R>
__global__ __launch_bounds__(256) void dw3x3_fwd_bn_kernel(const float* __restrict__ ze, const float* __restrict__ wgt, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, const float* __restrict__ center,
                                                           float* __restrict__ zd, float* __restrict__ part, DwGeom g, int bands, int tiles_per_band) {
    constexpr int TH = 4, RS = R == 2 ? 2 : 1, NR = RS == 2 ? TH + 2 : TH + 2 * R;
    constexpr int NQ = 8;
    __shared__ float s_acc[256][NQ + 1];
    const int tx = blockIdx.x % g.tiles_x, band = (blockIdx.x / g.tiles_x) % bands, b = blockIdx.x / (g.tiles_x * bands);
    const int flat = tx * 256 + threadIdx.x;
    const int ox = flat / g.CG, cg = flat - ox * g.CG;
    const bool live = ox < g.Wo;
    const int c0 = cg * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = zero4, s2 = zero4;
    if (live) {
        const float* xb = ze + (int64_t)b * g.H * g.W * g.C + c0;
        float* yb = zd + (int64_t)b * g.Ho * g.Wo * g.C + c0;
        int ixc[3];
        bool okx[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ix = ox - g.pl + j * R;
            okx[j] = ix >= 0 && ix < g.W;
            ixc[j] = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
        }
        float4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = ld4(wgt + k * g.C + c0);
        const float4 sc = ld4(scale + c0), sh = ld4(shift + c0), ctr = center ? ld4(center + c0) : zero4;
        const float lo = act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
        const int ty_end = (band + 1) * tiles_per_band < g.tiles_y ? (band + 1) * tiles_per_band : g.tiles_y;
        for (int ty = band * tiles_per_band; ty < ty_end; ++ty) {
            const int oy0 = (ty / RS) * (TH * RS) + (ty % RS);
            const int iy0 = oy0 - g.pt;
            float4 in[NR][3];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int iy = iy0 + rr * (RS == 2 ? R : 1);
                const bool oky = iy >= 0 && iy < g.H;
                const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
                const float* rp = xb + (int64_t)iyc * g.W * g.C;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 y = muladd4_pk(ld4(rp + (int64_t)ixc[j] * g.C), sc, sh);
                    const bool ok = oky && okx[j];                    // SAME padding pads the ACTIVATION with zeros
                    in[rr][j] = make_float4(ok ? __builtin_amdgcn_fmed3f(y.x, lo, hi) : 0.f, ok ? __builtin_amdgcn_fmed3f(y.y, lo, hi) : 0.f,
                                            ok ? __builtin_amdgcn_fmed3f(y.z, lo, hi) : 0.f, ok ? __builtin_amdgcn_fmed3f(y.w, lo, hi) : 0.f);
                }
            }
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int oy = oy0 + r * RS;
                if (oy >= g.Ho) break;
                float4 acc = zero4;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float4 v = in[RS == 2 ? r + i : r + i * R][j];
                        const float4 w4 = wv[i * 3 + j];
                        acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                        acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
                    }
                st4(yb + ((int64_t)oy * g.Wo + ox) * g.C, acc);
                const float4 d = sub4_pk(acc, ctr);
                s1 = add4_pk(s1, d);
                s2 = add4_pk(s2, mul4_pk(d, d));
            }
        }
    }
    float* sa = s_acc[threadIdx.x];
    sa[0] = s1.x; sa[1] = s1.y; sa[2] = s1.z; sa[3] = s1.w; sa[4] = s2.x; sa[5] = s2.y; sa[6] = s2.z; sa[7] = s2.w;
    sa[NQ] = live ? 1.f : -1.f;
    __syncthreads();
    const int cg_first = (tx * 256) % g.CG;
    const int ngroups = g.CG < 256 ? g.CG : 256;
    float* row = part + (int64_t)blockIdx.x * 2 * g.C;                  // [sum (z - center) | sum (z - center)^2][C]
    for (int e = threadIdx.x; e < ngroups * NQ; e += 256) {
        const int gi = e / NQ, qn = e - gi * NQ;
        const int cgw = (cg_first + gi) % g.CG;
        float s = 0.f;
        for (int t = gi; t < 256; t += g.CG)                            // threads t with (tx * 256 + t) % CG == cgw, ascending
            if (s_acc[t][NQ] >= 0.f) s += s_acc[t][qn];
        row[(qn >> 2) * g.C + cgw * 4 + (qn & 3)] = s;
    }
}

size_t depthwise_fwd_bn_scratch(int B, int H, int W, int C, int rate) {
    DwGeom f; int bands, tpb;
    if (C % 4 != 0 || C / 4 > 256) return (size_t)-1;
    dw_walk_plan(B, H, W, C, rate, 1024, &f, &bands, &tpb);
    return (size_t)f.tiles_x * bands * B * 2 * C;
}

// zd [B,H,W,C] = dwconv(act(ze scale + shift), w) (stride 1, rate 1|2), partial rows [rows][2][C] of (sum (zd - center), sum (zd - center)^2)
int launch_depthwise_fwd_bn(const float* ze, int B, int H, int W, int C, const float* w, int rate, const float* scale, const float* shift, int act,
                            const float* center, float* zd, float* scratch, int* rows_out, hipStream_t st) {
    DwGeom f;
    int bands, tpb;
    AMS_REQUIRE(C % 4 == 0 && C / 4 <= 256 && (rate == 1 || rate == 2), "depthwise_fwd_bn: C=%d rate=%d", C, rate);
    dw_walk_plan(B, H, W, C, rate, 1024, &f, &bands, &tpb);
    const unsigned nb = (unsigned)f.tiles_x * bands * B;
    *rows_out = (int)nb;
    note_kernel(rate == 2 ? "dw3x3_fwd_bn_kernel<2>" : "dw3x3_fwd_bn_kernel<1>");
    if (rate == 1) hipLaunchKernelGGL((dw3x3_fwd_bn_kernel<1>), dim3(nb), dim3(256), 0, st, ze, w, scale, shift, act, center, zd, scratch, f, bands, tpb);
    else hipLaunchKernelGGL((dw3x3_fwd_bn_kernel<2>), dim3(nb), dim3(256), 0, st, ze, w, scale, shift, act, center, zd, scratch, f, bands, tpb);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// weight gradient: dw[i,j,c] = sum_{b,oy,ox} x[b, oy*S-pt+i*R, ox*S-pl+j*R, c] * dy[b,oy,ox,c]
// A thread owns 4 channels and walks work items = (image, group of TH = 4 output rows, output column): the
// (TH-1)*S + 2R + 1 input rows that the group touches are each loaded once (3 taps per row) and feed up to 3 output
// rows, so an item costs NU*3 + 4 vector loads for 4 pixels instead of 10 per pixel (the tap re-reads were saturating
// L2, not HBM).  All loads of an item are requested up front, branch-free (clamped address + select).  A block owns a
// contiguous range of items; its pixel slots are added through LDS in a fixed order and the block partials by
// launch_reduce_splits (deterministic).
template <int S, int R>
__global__ __launch_bounds__(256) void dw3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ part, DwGeom g, int64_t items_per_block) {
    constexpr int TH = 4;
    constexpr int NU = (TH - 1) * S + 2 * R + 1;                    // distinct input rows of an item
    extern __shared__ __attribute__((aligned(16))) float sred[];   // [slots][9][C]
    const int cg = threadIdx.x % g.CG, slot = threadIdx.x / g.CG;
    const int c0 = cg * 4;
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tiles_y = (g.Ho + TH - 1) / TH;
    const int64_t total = (int64_t)g.B * tiles_y * g.Wo;
    const int64_t p_begin = blockIdx.x * items_per_block;
    int64_t p_end = p_begin + items_per_block;
    if (p_end > total) p_end = total;
    if (slot < g.slots) {
        for (int64_t p = p_begin + slot; p < p_end; p += g.slots) {
            const int pi = (int)p;                                   // < 2^31 (checked on the host)
            const int ox = pi % g.Wo;
            const int t2 = pi / g.Wo;
            const int ty = t2 % tiles_y, b = t2 / tiles_y;
            const int oy0 = ty * TH;
            float4 d[TH];
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int oy = oy0 + r;
                const int oyc = oy < g.Ho ? oy : g.Ho - 1;
                const float4 v = ld4(dy + (((int64_t)b * g.Ho + oyc) * g.Wo + ox) * g.C + c0);
                const bool ok = oy < g.Ho;
                d[r] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
            const float* xb = x + (int64_t)b * g.H * g.W * g.C + c0;
            float4 xv[NU][3];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int iy = oy0 * S - g.pt + u;
                const bool oky = iy >= 0 && iy < g.H;
                const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ix = ox * S - g.pl + j * R;
                    const bool ok = oky && ix >= 0 && ix < g.W;
                    const int ixc = ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix);
                    const float4 v = ld4(xb + ((int64_t)iyc * g.W + ixc) * g.C);
                    xv[u][j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    // input row u is tap row i of output row r iff u = r*S + i*R
                    if ((u - i * R) >= 0 && (u - i * R) % S == 0 && (u - i * R) / S < TH) {
                        const float4 dd = d[(u - i * R) / S];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            float4& a = acc[i * 3 + j];
                            a.x = fmaf(xv[u][j].x, dd.x, a.x); a.y = fmaf(xv[u][j].y, dd.y, a.y);
                            a.z = fmaf(xv[u][j].z, dd.z, a.z); a.w = fmaf(xv[u][j].w, dd.w, a.w);
                        }
                    }
                }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) st4(sred + ((int64_t)slot * 9 + k) * g.C + c0, acc[k]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * g.C; e += blockDim.x) {
        float s = 0.f;
        for (int sl = 0; sl < g.slots; ++sl) s += sred[(int64_t)sl * 9 * g.C + e];
        part[(int64_t)blockIdx.x * 9 * g.C + e] = s;
    }
}

// blocks of the depthwise weight-gradient pass: ~8 pixels per thread, at most 1024 blocks (the partials are reduced by
// a second kernel).  slots = pixel columns per block (256 / (C/4)): wide layers have ONE, so the block count must not be
// derived from a fixed pixels-per-block figure (that left the 960-channel layers with 68 blocks on 256 CUs).
static int dw_wgrad_blocks(int64_t total_items, int C) {
    const int cg = C / 4;
    const int slots = 256 / cg < 1 ? 1 : 256 / cg;
    int64_t blocks = cdiv64(total_items, (int64_t)slots * 2);      // ~2 items (8 pixels) per thread
    // every block writes a 9 x C partial that the second stage reads back: keep that below ~1/4 of the x + dy traffic
    // (4 pixels per item, 2 tensors: partial bytes <= items * 4 * 2 * C / 4  ->  blocks <= items * 2 / 9)
    const int64_t by_traffic = total_items * 2 / 9 > 1 ? total_items * 2 / 9 : 1;
    if (blocks > by_traffic) blocks = by_traffic;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

size_t depthwise_wgrad_scratch(int B, int H, int W, int C, int stride, int rate) {
    int Ho, Wo, p;
    same_pad(H, 3, stride, rate, &Ho, &p);
    same_pad(W, 3, stride, rate, &Wo, &p);
    return (size_t)dw_wgrad_blocks((int64_t)B * ((Ho + 3) / 4) * Wo, C) * 9 * C;
}

int launch_depthwise_wgrad(const float* x, const float* dy, int B, int H, int W, int C, int stride, int rate,
                           float* dw, float* scratch, size_t scratch_floats, hipStream_t st) {
    DwGeom g;
    int rc = dw_geom(B, H, W, C, stride, rate, false, &g);
    if (rc) return rc;
    const int64_t total = (int64_t)B * ((g.Ho + 3) / 4) * g.Wo;      // work items: 4 output rows x 1 column
    int blocks = dw_wgrad_blocks(total, C);
    AMS_REQUIRE(scratch_floats >= (size_t)blocks * 9 * C, "depthwise wgrad: scratch too small");
    AMS_REQUIRE(total < 0x7fffffffLL, "depthwise wgrad: too many pixels");
    const int threads = g.CG * g.slots;
    const size_t lds = (size_t)g.slots * 9 * C * sizeof(float);
    AMS_REQUIRE(lds <= 64 * 1024, "depthwise wgrad: LDS %zu too large", lds);
    // the items are dealt out up front, so a grid a little larger than what is co-resident costs a whole second round:
    // never launch more blocks than fit at once
    const void* fn = stride == 2 ? (const void*)dw3x3_wgrad_kernel<2, 1> : rate == 2 ? (const void*)dw3x3_wgrad_kernel<1, 2>
                                                                                   : (const void*)dw3x3_wgrad_kernel<1, 1>;
    int per_cu = 1, cus = 256;
    RUN_RC(func_blocks_per_cu(fn, threads, lds, &per_cu));
    RUN_RC(device_cus(&cus));
    if (blocks > cus * per_cu) blocks = cus * per_cu;
    const int64_t ppb = cdiv64(total, blocks);
    note_kernel(stride == 2 ? "dw3x3_wgrad_kernel<2, 1>" : rate == 2 ? "dw3x3_wgrad_kernel<1, 2>" : "dw3x3_wgrad_kernel<1, 1>");
    if (stride == 1 && rate == 1)
        hipLaunchKernelGGL((dw3x3_wgrad_kernel<1, 1>), dim3(blocks), dim3(threads), lds, st, x, dy, scratch, g, ppb);
    else if (stride == 2)
        hipLaunchKernelGGL((dw3x3_wgrad_kernel<2, 1>), dim3(blocks), dim3(threads), lds, st, x, dy, scratch, g, ppb);
    else
        hipLaunchKernelGGL((dw3x3_wgrad_kernel<1, 2>), dim3(blocks), dim3(threads), lds, st, x, dy, scratch, g, ppb);
    AMS_CHECK_LAUNCH();
    return launch_reduce_splits(scratch, blocks, (int64_t)9 * C, dw, st);
}

}  // namespace ams

// One whole inverted-residual block of the early (high-resolution) section in one kernel, frozen inference:
//   expand 1x1 + BN + ReLU6  ->  depthwise 3x3 (stride 1 | 2) + BN + ReLU6  ->  project 1x1 + BN (+ block input)
// Only the block input and the block output touch HBM: neither the 6x-expanded tensor nor the depthwise result `d` is ever
// written (SURVEY §8 d4 "block-fused convention").  In the layer-by-layer plan `d` alone is 1.6 GB of traffic per 32 frames over
// blocks 1-6, and the five early project GEMMs that read it back cost 0.47 ms of a 4.5 ms step.
//
// Every WAVE owns a TH x TW tile of OUTPUT pixels (8 x 8 at stride 1, 4 x 8 at stride 2) and walks the expanded channels 16 at a time:
//   phase 1  exact-f32 MFMA GEMM over the tile's input pixels incl. the 3x3 halo (fragments gathered once per tile and kept in
//            registers for all chunks), BN + ReLU6, into the wave's PRIVATE LDS tile (zeros outside the image = the depthwise conv's
//            SAME padding);
//   phase 2  depthwise 3x3 from that tile, BN + ReLU6, lane (pixel l15, k-group q) forming the four channels n0 + 4q .. +3 of its
//            pixel: exactly the lane's operand of
//   phase 3  the project MFMAs (exact f32, roles swapped: a lane owns 4 consecutive output channels of one pixel), whose
//            accumulators live across the chunks: `d` never leaves the registers.
// The weights of a chunk are MFMA operand A and live in registers too (4*KC + 4*NTO floats per lane, requested one chunk ahead
// straight from L2).  There is ONE block barrier (after the per-channel vectors of all chunks are staged); after it the four waves
// of a block run free, so one wave's MFMA-heavy expand phase overlaps another's VALU/LDS-heavy depthwise phase on the same SIMD —
// the tile-per-block forms of this kernel (barriers around every phase) spent 46 % of their wave cycles waiting
// (profiles/r02_block_kernel_sq.txt).
// After the last chunk: BN of the project layer, residual (= the block input at the output pixel), float4 stores.
// Products, k order and tap order are those of the kernels it replaces (pw_gemm_f32_s walks k in 16-wide chunks, 4 MFMA steps
// each, k = 16c + 4q + j; so does phase 3 across the channel chunks): the result is bit-identical to the layer-by-layer plan.
#include <type_traits>

#include "pw_common.hpp"
#include "split_bf16.hpp"

namespace ams {

struct BlkArgs {
    const float* x;          // [B, H, W, Cin]  block input (also the residual operand)
    int B, H, W, Cin;
    const float* w_exp;      // [Cin, Cexp]
    const float* sc_e; const float* sh_e;
    int Cexp;
    const float* w_dw;       // [9, Cexp]
    const float* sc_d; const float* sh_d;
    const float* w_pj;       // [Cexp, Cout]
    const float* sc_p; const float* sh_p;
    int Cout;
    int act_e, act_d, act_p, residual;
    float* y;                // [B, Ho, Wo, Cout]
    int Ho, Wo, pt, pl;
    int tiles_x, tiles_y, chunks;
    const float* vecs;       // optional: sc_e | sh_e | sc_d | sh_d | w_dw[9] already packed as [13][Cexp] (the engine packs at freeze)
    // STEM form (first block of the network): x = the frames [B, fH, fW, 3] (uint8 or float); the "expand" layer is the stem conv
    // (pad 127.5, x * ps - 1, dense 3x3 stride 2: K = 27 taps gathered per position), H x W = the stem's output grid
    int fH, fW, spt, spl;
    float ps;
    // X6 form (Cin 24 / 32): the expand products as six bf16 MFMAs on three-part splits (f32-level, cf. k_pw_x3.hip) instead of eight
    // exact-f32 MFMAs of twice the latency: wparts = the expand layer's panels [part][Cexp][32] bf16 (k contiguous, zero-padded),
    // parts `wplane` elements apart
    const uint16_t* wparts;
    int64_t wplane;
    // H16 form (AMS_MATMUL_SPLIT_F16; split_bf16.hpp): expand AND project products on two fp16 parts, 3 MFMAs each (16x16x16 for K = 16, 16x16x32
    // for K = 24 / 32, 16x16x16 per 16-channel chunk of the project layer) instead of 4 / 8 / 4 exact-f32 MFMAs of twice the latency:
    // he = the expand layer's fp16 panels [part][Cexp][32], hp = the project layer's [part][Cout][kp_p] (k contiguous, zero-padded)
    const uint16_t* he; int64_t he_plane;
    const uint16_t* hp; int64_t hp_plane; int kp_p;
    int timed;               // tools/ only (AMS_BLK_TIMED=1): sum shader-clock cycles per phase into g_blk_cycles (ams_debug_phase_cycles(1, ..))
};

// tools/ only: [0] prologue (vectors, input fragments, first weights), [1] expand phases, [2] depthwise + project phases, [3] epilogue, [6] waves
// (1024 rows, a wave adds to row (4 block + wave) mod 1024: one shared row serialises 10^5 atomics and slows the kernel 7x)
__device__ unsigned long long g_blk_cycles[1024][8];

// value of the normalised, 127.5-padded frame at (iy, ix, ch) in padded coordinates; outside of it the stem's SAME zero padding
template <typename TIn>
__device__ __forceinline__ float blk_frame_value(const TIn* img, int H, int W, int iy, int ix, int ch, float ps) {
    const bool inside = iy >= 0 && ix >= 0 && iy <= H && ix <= W;
    const bool pad = iy >= H || ix >= W;
    const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const int ixc = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    float raw = (float)img[((int64_t)iyc * W + ixc) * 3 + ch];
    raw = pad ? 127.5f : raw;
    const float v = __fsub_rn(__fmul_rn(raw, ps), 1.0f);
    return inside ? v : 0.f;
}

// TIn = void: the block input is an f32 activation tensor; uint8_t / float: STEM form, the input is the frame batch
constexpr int blk_act_pitch(int s, int tw) { return s == 1 && tw == 16 ? 24 : 20; }

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mma16_f16(const u32x2_t& a, const u32x2_t& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}

// HP (H16 only): the project products on fp16 parts too; false = exact-f32 project MFMAs behind an fp16 expand
template <int S, int KC, int NTO, int TH, int TW, typename TIn = void, bool X6 = false, bool H16 = false, bool HP = H16>
__global__ __launch_bounds__(256, 2) void block_kernel(BlkArgs a, unsigned ntiles) {
    static_assert(!HP || H16, "HP is a refinement of H16");
    constexpr bool PKDW = S == 1 || TH == 2;          // packed f32 in the depthwise phase (see there)
    constexpr bool SLIDE = S == 1 && TW == 16 && TH % 2 == 0;      // the depthwise taps slide down the lane's column (see there)
    constexpr bool STEM = !std::is_void<TIn>::value;
    static_assert(!(X6 && STEM) && !(H16 && STEM) && !(X6 && H16), "the stem stays exact f32; one split form at a time");
    constexpr bool H16W = H16 && KC == 2;             // expand fragments of 8 k per lane (16x16x32); KC == 1: 4 k per lane (16x16x16)
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int NPIX = IH * IW;
    constexpr int NRG = (NPIX + 15) / 16;             // 16-pixel row groups of the wave's input tile (halo included)
    // pitch of a 16-channel row of the wave's tile.  With 16 consecutive positions per row group (4 x 16 tile) 96 bytes are conflict-free
    // for the tap reads and the expand stores under the real lane groups of ds_read_b128 / ds_write_b128; 80 bytes are 2-way conflicted on
    // three of sixteen lanes, and so is every pitch for the other tile shapes (brute-forced)
    constexpr int AP = blk_act_pitch(S, TW);
    constexpr int MRO = TH * TW / 16;                 // output row groups of the wave's tile
    static_assert(TH * TW % 16 == 0, "whole output row groups");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // per-channel vectors of ALL chunks, staged once per block: sc_e | sh_e | sc_d | sh_d | depthwise taps [9][Cexp]
    float* sVec = smem;                               // [13][Cexp]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    float* sAct = smem + 13 * a.Cexp + wave * (NRG * 16 * AP);       // this wave's expanded tile, 16 channels at a time
    unsigned long long tc[4] = {0, 0, 0, 0}, tl_ = a.timed ? __builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int slot) {
        if (a.timed) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tc[slot] += now - tl_;
            tl_ = now;
        }
    };

    // ---- every WAVE owns a TH x TW tile of output pixels: no block barrier after the prologue, the four waves of a block
    // drift apart and one wave's MFMA-heavy expand phase overlaps another's VALU/LDS-heavy depthwise phase on the same SIMD
    if (a.vecs) {                                      // block-uniform: one round trip of float4 copies
        for (int e = tid; e < 13 * a.Cexp / 4; e += 256) st4(sVec + 4 * e, ld4(a.vecs + 4 * e));
    } else {
        for (int e = tid; e < 13 * a.Cexp; e += 256) {
            const int which = e / a.Cexp, c = e - which * a.Cexp;
            const float* src = which == 0 ? a.sc_e : which == 1 ? a.sh_e : which == 2 ? a.sc_d : which == 3 ? a.sh_d : a.w_dw + (which - 4) * a.Cexp;
            sVec[e] = src[c];
        }
    }
    unsigned wt = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    const bool live = wt < ntiles;
    if (!live) wt = ntiles - 1;                        // surplus waves of the last block repeat the last tile and store nothing
    const int tx = wt % a.tiles_x;
    unsigned t1 = wt / a.tiles_x;
    const int ty = t1 % a.tiles_y;
    const int b = t1 / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;

    // BN + activation = one fma and one v_med3_f32 per value: clamp to [lo, hi] with wave-uniform bounds; an input-tile position
    // outside the image gets hi = lo = 0 (the depthwise conv's SAME padding) instead of four selects
    const float lo_e = a.act_e == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_e = a.act_e == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    const float lo_d = a.act_d == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_d = a.act_d == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    // ---- the wave's input fragments, requested up front and kept for all chunks (clamped addresses, branch-free)
    const float* xb = STEM ? nullptr : a.x + (int64_t)b * a.H * a.W * a.Cin;
    float4 areg[(X6 || H16) ? 1 : NRG][KC];
    bf16x8 xp[X6 ? NRG : 1][3];                        // X6: the fragments as bf16 parts (k = 8q .. 8q + 7 of the lane's pixel), split once
    u32x4 xh8[H16W ? NRG : 1], xl8[H16W ? NRG : 1];    // H16, K = 24 / 32: the fragments as fp16 parts (k = 8q .. 8q + 7)
    u32x2_t xh4[(H16 && !H16W) ? NRG : 1], xl4[(H16 && !H16W) ? NRG : 1];      // H16, K = 16: k = 4q .. 4q + 3
    unsigned inside_mask = 0;
    if constexpr (STEM) {
        // this lane's taps: k = 16c + 4q + j -> (dy, dx, channel) of the 3x3x3 receptive field, k >= 27 padding (value 0)
        typedef typename std::conditional<STEM, TIn, float>::type TI;
        const TI* img = reinterpret_cast<const TI*>(a.x) + (int64_t)b * a.fH * a.fW * 3;
        int toff[KC][4], tdy[KC][4], tdx[KC][4], tch[KC][4];
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * c + 4 * q + j, tap = k / 3;
                tch[c][j] = k < 27 ? k - tap * 3 : -1;
                tdy[c][j] = tap / 3;
                tdx[c][j] = tap - tdy[c][j] * 3;
                toff[c][j] = k < 27 ? (tdy[c][j] * a.fW + tdx[c][j]) * 3 + tch[c][j] : 0;
            }
        // interior tiles: every stem position of the halo tile exists and every tap lies inside the frame (wave-uniform)
        const bool interior = iy0 >= 0 && iy0 + IH <= a.H && ix0 >= 0 && ix0 + IW <= a.W && iy0 * 2 - a.spt >= 0 &&
                              (iy0 + IH - 1) * 2 - a.spt + 2 <= a.fH - 1 && ix0 * 2 - a.spl >= 0 && (ix0 + IW - 1) * 2 - a.spl + 2 <= a.fW - 1;
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
            int m = rg * 16 + l15;
            const bool in_tile = m < NPIX;
            if (m > NPIX - 1) m = NPIX - 1;
            const int ty_i = m / IW, tx_i = m - ty_i * IW;
            const int sy = iy0 + ty_i, sx = ix0 + tx_i;                  // position in the stem's output grid
            if (in_tile && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) inside_mask |= 1u << rg;
            const int fy = sy * 2 - a.spt, fx = sx * 2 - a.spl;
            float t[KC][4];
            if (interior) {
                const TI* p0 = img + ((int64_t)fy * a.fW + fx) * 3;
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[c][j] = __fsub_rn(__fmul_rn((float)p0[toff[c][j]], a.ps), 1.0f);
            } else {
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        t[c][j] = blk_frame_value(img, a.fH, a.fW, fy + tdy[c][j], fx + tdx[c][j], tch[c][j] < 0 ? 0 : tch[c][j], a.ps);
            }
#pragma unroll
            for (int c = 0; c < KC; ++c)
                areg[rg][c] = make_float4(tch[c][0] >= 0 ? t[c][0] : 0.f, tch[c][1] >= 0 ? t[c][1] : 0.f, tch[c][2] >= 0 ? t[c][2] : 0.f,
                                          tch[c][3] >= 0 ? t[c][3] : 0.f);
        }
    } else {
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
        const int m = rg * 16 + l15;
        const int ty_i = m / IW, tx_i = m - ty_i * IW;
        const int iy = iy0 + ty_i, ix = ix0 + tx_i;
        if (m < NPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) inside_mask |= 1u << rg;
        const int iyc = iy < 0 ? 0 : (iy > a.H - 1 ? a.H - 1 : iy), ixc = ix < 0 ? 0 : (ix > a.W - 1 ? a.W - 1 : ix);
        const float* px = xb + ((int64_t)iyc * a.W + ixc) * a.Cin;
        if constexpr (X6 || H16W) {
            int k0 = 8 * q, k1 = 8 * q + 4;
            const bool ok0 = k0 < a.Cin, ok1 = k1 < a.Cin;
            if (k0 > a.Cin - 4) k0 = a.Cin - 4;
            if (k1 > a.Cin - 4) k1 = a.Cin - 4;
            float4 u = ld4(px + k0), v = ld4(px + k1);
            u = make_float4(ok0 ? u.x : 0.f, ok0 ? u.y : 0.f, ok0 ? u.z : 0.f, ok0 ? u.w : 0.f);
            v = make_float4(ok1 ? v.x : 0.f, ok1 ? v.y : 0.f, ok1 ? v.z : 0.f, ok1 ? v.w : 0.f);
            if constexpr (X6) split8(u, v, xp[rg][0], xp[rg][1], xp[rg][2]);
            else {
                f16x8 h, l;
                split8_f16(u, v, h, l);
                xh8[rg] = __builtin_bit_cast(u32x4, h); xl8[rg] = __builtin_bit_cast(u32x4, l);
            }
        } else if constexpr (H16) {
            int koff = 4 * q;
            const bool ok = koff < a.Cin;
            if (koff > a.Cin - 4) koff = a.Cin - 4;
            const float4 v = ld4(px + koff);
            unsigned h[2], l[2];
            split4_f16(make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f), h, l);
            xh4[rg] = (u32x2_t){h[0], h[1]}; xl4[rg] = (u32x2_t){l[0], l[1]};
        } else {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                int koff = c * 16 + 4 * q;
                const bool ok = koff < a.Cin;
                if (koff > a.Cin - 4) koff = a.Cin - 4;
                const float4 v = ld4(px + koff);
                areg[rg][c] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        }
    }
    }
    // ---- chunk weights in registers (weights = MFMA operand A: a lane needs w[k = 16c + 4q + j][n = l15] only): pointers are set
    // up once and bumped per chunk, the next chunk's values are requested as soon as the current ones have been consumed
    const float* pwa[KC][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * c + 4 * q + j;
            pwa[c][j] = a.w_exp + (int64_t)(k < a.Cin ? k : a.Cin - 1) * a.Cexp + l15;
        }
    const float* pwp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pwp[j] = a.w_pj + (int64_t)(4 * q + j) * a.Cout + (l15 < a.Cout ? l15 : 0);
    int ncl[NTO];                                      // column offset of output tile t, clamped into the row
#pragma unroll
    for (int t = 0; t < NTO; ++t) ncl[t] = 16 * t + l15 < a.Cout ? 16 * t : 0;
    float wa[(X6 || H16) ? 1 : KC][4], wp[HP ? 1 : 4][NTO];
    bf16x8 wq[3];                                      // X6: this lane's expand-weight fragments (channel n0 + l15, k = 8q .. 8q + 7), three parts
    const uint16_t* pwq = X6 ? a.wparts + (int64_t)l15 * 32 + 8 * q : nullptr;
    u32x4 wh8, wl8;                                    // H16: the same as fp16 parts (K = 24 / 32) ...
    u32x2_t wh4, wl4;                                  // ... or k = 4q .. 4q + 3 (K = 16)
    const uint16_t* pwh = H16 ? a.he + (int64_t)l15 * 32 + (H16W ? 8 : 4) * q : nullptr;
    u32x2_t ph[HP ? NTO : 1], pl[HP ? NTO : 1];        // HP: project-weight fragments of the chunk (column 16t + l15, k = n0 + 4q .. + 3)
    const uint16_t* pwj[HP ? NTO : 1];
    if constexpr (HP) {
#pragma unroll
        for (int t = 0; t < NTO; ++t) pwj[t] = a.hp + (int64_t)(16 * t + l15 < a.Cout ? 16 * t + l15 : 0) * a.kp_p + 4 * q;
    }
    auto load_wa = [&](int n0) {
        if constexpr (X6) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) wq[pp] = *reinterpret_cast<const bf16x8*>(pwq + pp * a.wplane + (int64_t)n0 * 32);
        } else if constexpr (H16W) {
            wh8 = *reinterpret_cast<const u32x4*>(pwh + (int64_t)n0 * 32);
            wl8 = *reinterpret_cast<const u32x4*>(pwh + a.he_plane + (int64_t)n0 * 32);
        } else if constexpr (H16) {
            wh4 = *reinterpret_cast<const u32x2_t*>(pwh + (int64_t)n0 * 32);
            wl4 = *reinterpret_cast<const u32x2_t*>(pwh + a.he_plane + (int64_t)n0 * 32);
        } else {
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) wa[c][j] = pwa[c][j][n0];
        }
    };
    auto load_wp = [&](int n0) {
        if constexpr (HP) {
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                ph[t] = *reinterpret_cast<const u32x2_t*>(pwj[t] + n0);
                pl[t] = *reinterpret_cast<const u32x2_t*>(pwj[t] + a.hp_plane + n0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NTO; ++t) wp[j][t] = pwp[j][(int64_t)n0 * a.Cout + ncl[t]];
        }
    };
    load_wa(0);
    load_wp(0);

    f32x4 out[MRO][NTO], outx[HP ? MRO : 1][HP ? NTO : 1];        // HP: the cross terms of the project products
#pragma unroll
    for (int i = 0; i < MRO; ++i)
#pragma unroll
        for (int t = 0; t < NTO; ++t) out[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (HP ? MRO : 1); ++i)
#pragma unroll
        for (int t = 0; t < (HP ? NTO : 1); ++t) outx[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                   // sVec is staged: the only block barrier of the kernel
    lap(0);

    const int chunks = a.Cexp / 16;
    for (int ci = 0; ci < chunks; ++ci) {
        const int n0 = ci * 16;
        const int n_next = ci + 1 < chunks ? n0 + 16 : n0;          // the last chunk re-requests itself: no branch around the loads
        // ---- phase 1: expand GEMM over the wave's input tile, BN + ReLU6, into the wave's LDS tile (zeros outside the image)
        {
            const float4 sc = ld4(sVec + n0 + 4 * q), sh = ld4(sVec + a.Cexp + n0 + 4 * q);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (X6) {                    // the six products of k_pw_x3.hip, smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[2], xp[rg][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0], xp[rg][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[1], xp[rg][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[1], xp[rg][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0], xp[rg][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0], xp[rg][0], acc, 0, 0, 0);
                } else if constexpr (H16) {            // cross terms (wl xh, wh xl) in their own accumulator, then the main term
                    f32x4 accx = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (H16W) {
                        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl8), __builtin_bit_cast(f16x8, xh8[rg]), accx, 0, 0, 0);
                        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh8), __builtin_bit_cast(f16x8, xl8[rg]), accx, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh8), __builtin_bit_cast(f16x8, xh8[rg]), acc, 0, 0, 0);
                    } else {
                        accx = mma16_f16(wl4, xh4[rg], accx);
                        accx = mma16_f16(wh4, xl4[rg], accx);
                        acc = mma16_f16(wh4, xh4[rg], acc);
                    }
                    acc = combine_f16(acc, accx);
                } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][0], areg[rg][c].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][1], areg[rg][c].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][2], areg[rg][c].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][3], areg[rg][c].w, acc, 0, 0, 0);
                    }
                }
                const bool inside = (inside_mask >> rg) & 1u;
                const float lo = inside ? lo_e : 0.f, hi = inside ? hi_e : 0.f;
                float4 v;
                const float4 bn = muladd4_pk(make_float4(acc[0], acc[1], acc[2], acc[3]), sc, sh);      // v_pk_mul + v_pk_add: two roundings
                v.x = __builtin_amdgcn_fmed3f(bn.x, lo, hi);
                v.y = __builtin_amdgcn_fmed3f(bn.y, lo, hi);
                v.z = __builtin_amdgcn_fmed3f(bn.z, lo, hi);
                v.w = __builtin_amdgcn_fmed3f(bn.w, lo, hi);
                st4(sAct + (rg * 16 + l15) * AP + 4 * q, v);
            }
        }
        load_wa(n_next);                               // in flight across the depthwise / project phase
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lap(1);

        // ---- phases 2 + 3 per 16-pixel output row group: lane (l15, q) forms the depthwise result of pixel l15 for the channels
        // n0 + 4q .. +3 — exactly its operand of the project MFMAs (k = n0 + 4q + j), so `d` never leaves the registers
        {
            float4 w9[9];                                  // depthwise weights of these 4 channels, held for the chunk's row groups
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) w9[tp] = ld4(sVec + (4 + tp) * a.Cexp + n0 + 4 * q);
            const float4 scd = ld4(sVec + 2 * a.Cexp + n0 + 4 * q), shd = ld4(sVec + 3 * a.Cexp + n0 + 4 * q);
            // BN + activation of a finished depthwise accumulator, then the lane's share of the project MFMAs of row group i
            auto finish = [&](int i, const float4& acc) {
                float dv[4];
                const float4 bn = PKDW ? muladd4_pk(acc, scd, shd)
                                       : make_float4(acc.x * scd.x + shd.x, acc.y * scd.y + shd.y, acc.z * scd.z + shd.z, acc.w * scd.w + shd.w);
                dv[0] = __builtin_amdgcn_fmed3f(bn.x, lo_d, hi_d); dv[1] = __builtin_amdgcn_fmed3f(bn.y, lo_d, hi_d);
                dv[2] = __builtin_amdgcn_fmed3f(bn.z, lo_d, hi_d); dv[3] = __builtin_amdgcn_fmed3f(bn.w, lo_d, hi_d);
                if constexpr (HP) {
                    unsigned h[2], l[2];
                    split4_f16(make_float4(dv[0], dv[1], dv[2], dv[3]), h, l);
                    const u32x2_t dh = {h[0], h[1]}, dl = {l[0], l[1]};
#pragma unroll
                    for (int t = 0; t < NTO; ++t) outx[i][t] = mma16_f16(pl[t], dh, outx[i][t]);
#pragma unroll
                    for (int t = 0; t < NTO; ++t) outx[i][t] = mma16_f16(ph[t], dl, outx[i][t]);
#pragma unroll
                    for (int t = 0; t < NTO; ++t) out[i][t] = mma16_f16(ph[t], dh, out[i][t]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < NTO; ++t) out[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[j][t], dv[j], out[i][t], 0, 0, 0);
                }
            };
            if constexpr (SLIDE) {
                // stride 1, 16-pixel tile rows: row group i IS tile row i and the lane keeps its column, so the taps slide down the column — (TH + 2) x 3
                // LDS reads per chunk instead of TH x 9 (the tap reads were the largest item on the LDS pipe, which bounds these kernels: per chunk
                // and wave 47 b128 reads + 7 writes against 2 x 16 MFMA-pipe passes), two input rows per round trip.  A row group's FMAs keep their
                // order (tap rows outer, columns inner): same bits
                const float* col = sAct + l15 * AP + 4 * q;
                float4 acc[MRO];
#pragma unroll
                for (int i = 0; i < MRO; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u0 = 0; u0 < TH + 2; u0 += 2) {
                    float4 r[2][3];
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                        for (int jj = 0; jj < 3; ++jj) r[uu][jj] = ld4(col + ((u0 + uu) * IW + jj) * AP);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu) {
                        const int u = u0 + uu;                 // input row u is tap row ii of tile row u - ii
#pragma unroll
                        for (int ii = 0; ii < 3; ++ii) {
                            const int i = u - ii;
                            if (i >= 0 && i < MRO) {
#pragma unroll
                                for (int jj = 0; jj < 3; ++jj) fma4_pk(acc[i], r[uu][jj], w9[ii * 3 + jj]);
                            }
                        }
                        if (u >= 2) finish(u - 2, acc[u - 2]);
                    }
                }
            } else {
#pragma unroll
            for (int i = 0; i < MRO; ++i) {
                const int P = i * 16 + l15;
                const int ly = P / TW, lx = P - ly * TW;
                const float* tap0 = sAct + ((ly * S) * IW + lx * S) * AP + 4 * q;
                // all nine taps requested before the first is used: one LDS round trip per row group (left to itself under a
                // register bound, hipcc waits for every tap separately: 9 x ~100 cycles against 36 FMAs)
                float4 tp9[9];
#pragma unroll
                for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) tp9[ii * 3 + jj] = ld4(tap0 + (ii * IW + jj) * AP);
                asm volatile("" ::: "memory");
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    // packed f32 halves the issue slots of the depthwise phase; the 4 x 8 stride-2 tile of the split form has no
                    // registers left for the aligned pairs (60 spilled VGPRs, 159 -> 254 us): scalar there
                    if (PKDW) fma4_pk(acc, tp9[k], w9[k]);
                    else {
                        acc.x = fmaf(tp9[k].x, w9[k].x, acc.x); acc.y = fmaf(tp9[k].y, w9[k].y, acc.y);
                        acc.z = fmaf(tp9[k].z, w9[k].z, acc.z); acc.w = fmaf(tp9[k].w, w9[k].w, acc.w);
                    }
                }
                finish(i, acc);
            }
            }
        }
        load_wp(n_next);                               // in flight across the next chunk's expand phase
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lap(2);
    }

    // ---- project epilogue: BN, residual = block input at the output pixel (stride 1, Cin == Cout), 16-byte stores.  Coefficients and residual
    // values of ALL the wave's row groups are requested before the first store: y may alias x for all hipcc knows, so a load placed behind a
    // store waits for its own round trip (measured: 7-10 k cycles of a residual block's 45-63 k per wave in this epilogue, one round trip per
    // row group and column tile)
    float* yb = a.y + (int64_t)b * a.Ho * a.Wo * a.Cout;
    float4 scp[NTO], shp[NTO], rres[(!STEM) ? MRO : 1][(!STEM) ? NTO : 1];
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
        const int c4 = 16 * t + 4 * q < a.Cout ? 16 * t + 4 * q : 0;
        scp[t] = ld4(a.sc_p + c4);
        shp[t] = ld4(a.sh_p + c4);
    }
    if constexpr (!STEM) {
        if (a.residual) {                              // block-uniform
#pragma unroll
            for (int i = 0; i < MRO; ++i) {
                const int P = i * 16 + l15;
                const int ly = P / TW, lx = P - ly * TW;
                const int oy = oy0 + ly < a.Ho ? oy0 + ly : a.Ho - 1, ox = ox0 + lx < a.Wo ? ox0 + lx : a.Wo - 1;
#pragma unroll
                for (int t = 0; t < NTO; ++t) rres[i][t] = ld4(xb + ((int64_t)oy * a.W + ox) * a.Cin + (16 * t + 4 * q < a.Cout ? 16 * t + 4 * q : 0));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MRO; ++i) {
        const int P = i * 16 + l15;
        const int ly = P / TW, lx = P - ly * TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        if (!live || oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            const int c4 = 16 * t + 4 * q;
            if (c4 >= a.Cout) continue;
            float4 v;
            if constexpr (HP) out[i][t] = combine_f16(out[i][t], outx[HP ? i : 0][HP ? t : 0]);
            const float4 bn = muladd4_pk(make_float4(out[i][t][0], out[i][t][1], out[i][t][2], out[i][t][3]), scp[t], shp[t]);
            v.x = apply_act(bn.x, a.act_p); v.y = apply_act(bn.y, a.act_p);
            v.z = apply_act(bn.z, a.act_p); v.w = apply_act(bn.w, a.act_p);
            if constexpr (!STEM) {
                if (a.residual) {
                    const float4 r = rres[i][t];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
            }
            st4(yb + ((int64_t)oy * a.Wo + ox) * a.Cout + c4, v);
        }
    }
    if (a.timed) {
        lap(3);
        if (lane == 0) {
            unsigned long long* row = g_blk_cycles[(blockIdx.x * 4 + wave) & 1023];
            for (int i = 0; i < 4; ++i) atomicAdd(&row[i], tc[i]);
            atomicAdd(&row[6], 1ull);
        }
    }
}

template <int S, int KC, int NTO, int TH, int TW, typename TIn = void, bool X6 = false, bool H16 = false, bool HP = H16>
static int launch_blk_k(BlkArgs a, hipStream_t st) {
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int NRG = (IH * IW + 15) / 16;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = cdiv(a.Ho, TH);
    const size_t lds = ((size_t)13 * a.Cexp + (size_t)4 * NRG * 16 * blk_act_pitch(S, TW)) * sizeof(float);
    AMS_REQUIRE(lds <= 64 * 1024, "block kernel: needs %zu bytes of LDS", lds);
    const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.B;
    const int64_t nblocks = cdiv64(ntiles, 4);
    AMS_REQUIRE(ntiles > 0 && ntiles < 0x7fffffffLL, "block kernel: bad grid");
    static const std::string nm = "block_kernel<" + std::to_string(S) + ", " + std::to_string(KC) + ", " + std::to_string(NTO) + ", " +
                                  std::to_string(TH) + ", " + std::to_string(TW) +
                                  (std::is_void<TIn>::value ? ", void" : sizeof(typename std::conditional<std::is_void<TIn>::value, char, TIn>::type) == 1 ? ", unsigned char" : ", float") +
                                  (X6 ? ", true" : ", false") + (H16 ? ", true" : ", false") + (HP ? ", true>" : ", false>");      // as rocprofv3 prints the symbol
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((block_kernel<S, KC, NTO, TH, TW, TIn, X6, H16, HP>), dim3((unsigned)nblocks), dim3(256), lds, st, a, (unsigned)ntiles);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int S, int TH, int TW>
static int launch_blk_t(const BlkArgs& a, hipStream_t st) {
    const bool k1 = (a.Cin + 15) / 16 == 1, o2 = a.Cout <= 32;
    if (a.he) {
        // 64-wide project layers (four accumulator tiles per row group) keep the exact-f32 project MFMAs: measured 68 vs 81 us on the
        // 32 -> 192 -> 64 block (AMS_BLK_HP = 0 | 1 forces either)
        const bool hp = knobs().blk_hp >= 0 ? knobs().blk_hp != 0 : o2;
        if (k1) return o2 ? (hp ? launch_blk_k<S, 1, 2, TH, TW, void, false, true>(a, st) : launch_blk_k<S, 1, 2, TH, TW, void, false, true, false>(a, st))
                          : (hp ? launch_blk_k<S, 1, 4, TH, TW, void, false, true>(a, st) : launch_blk_k<S, 1, 4, TH, TW, void, false, true, false>(a, st));
        return o2 ? (hp ? launch_blk_k<S, 2, 2, TH, TW, void, false, true>(a, st) : launch_blk_k<S, 2, 2, TH, TW, void, false, true, false>(a, st))
                  : (hp ? launch_blk_k<S, 2, 4, TH, TW, void, false, true>(a, st) : launch_blk_k<S, 2, 4, TH, TW, void, false, true, false>(a, st));
    }
    if (k1) return o2 ? launch_blk_k<S, 1, 2, TH, TW>(a, st) : launch_blk_k<S, 1, 4, TH, TW>(a, st);
    if (a.wparts) return o2 ? launch_blk_k<S, 2, 2, TH, TW, void, true>(a, st) : launch_blk_k<S, 2, 4, TH, TW, void, true>(a, st);
    return o2 ? launch_blk_k<S, 2, 2, TH, TW>(a, st) : launch_blk_k<S, 2, 4, TH, TW>(a, st);
}

// sc_e | sh_e | sc_d | sh_d | w_dw[9][Cexp] -> one [13][Cexp] table (what a block stages into LDS with plain float4 copies)
__global__ void blk_pack_kernel(const float* sc_e, const float* sh_e, const float* sc_d, const float* sh_d, const float* w_dw, int Cexp, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 13 * Cexp) return;
    const int which = e / Cexp, c = e - which * Cexp;
    out[e] = which == 0 ? sc_e[c] : which == 1 ? sh_e[c] : which == 2 ? sc_d[c] : which == 3 ? sh_d[c] : w_dw[(which - 4) * Cexp + c];
}
int launch_block_pack(const float* sc_e, const float* sh_e, const float* sc_d, const float* sh_d, const float* w_dw, int Cexp, float* out, hipStream_t st) {
    hipLaunchKernelGGL(blk_pack_kernel, dim3(cdiv(13 * Cexp, 256)), dim3(256), 0, st, sc_e, sh_e, sc_d, sh_d, w_dw, Cexp, out);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// LDS of the tile-per-wave kernel: the [13][Cexp] table + four waves' expanded tiles (16 channels, halo included)
static size_t blk_lds_bytes(int stride, int th, int tw, int Cexp) {
    const int ih = (th - 1) * stride + 3, iw = (tw - 1) * stride + 3;
    const int nrg = (ih * iw + 15) / 16;
    return ((size_t)13 * Cexp + (size_t)4 * nrg * 16 * blk_act_pitch(stride, tw)) * sizeof(float);
}

// tile (output pixels per wave) for a block, or false when none of the instantiated tiles fits the 64 KB of LDS.
// Measured on the six early blocks at 32 frames (tools/blk_sweep.sh): 8 x 8 at stride 1 (4 x 16 in the X6 form: no spills, conflict-free tap
// reads along a row); at stride 2 4 x 8, or 2 x 8 when the project layer is 64 wide (four accumulator tiles per row group: the smaller tile
// keeps the registers of two waves per SIMD) — and whenever 4 x 8 does not fit (Cexp > 272)
static bool blk_pick_tile(int Cexp, int Cout, int stride, bool x6, int* th, int* tw) {
    if (stride == 1) {
        *th = x6 ? 4 : 8; *tw = x6 ? 16 : 8;
        if (blk_lds_bytes(1, *th, *tw, Cexp) <= 64 * 1024) return true;
        *th = 4; *tw = 8;
        return blk_lds_bytes(1, 4, 8, Cexp) <= 64 * 1024;
    }
    *th = Cout > 32 ? 2 : 4; *tw = 8;
    if (blk_lds_bytes(2, *th, *tw, Cexp) <= 64 * 1024) return true;
    *th = 2;
    return blk_lds_bytes(2, 2, 8, Cexp) <= 64 * 1024;
}

bool block_fused_supported(int Cin, int Cexp, int Cout, int stride, int rate, bool residual) {
    if (Cin % 4 != 0 || Cin > 32 || rate != 1 || (stride != 1 && stride != 2)) return false;
    if (Cout % 4 != 0 || Cout > 64) return false;
    if (residual && (stride != 1 || Cin != Cout)) return false;
    if (Cexp % 16 != 0 || Cexp > 384) return false;
    int th, tw;
    return blk_pick_tile(Cexp, Cout, stride, false, &th, &tw) && blk_pick_tile(Cexp, Cout, stride, true, &th, &tw);
}

int launch_block_fused(const float* x, int B, int H, int W, int Cin, const float* w_exp, const float* sc_e, const float* sh_e, int act_e, int Cexp,
                       const float* w_dw, int stride, const float* sc_d, const float* sh_d, int act_d, const float* w_pj, const float* sc_p,
                       const float* sh_p, int act_p, int Cout, bool residual, float* y, hipStream_t st, const float* vecs, const uint16_t* wparts,
                       int64_t wplane, const uint16_t* h_exp, int64_t h_exp_plane, const uint16_t* h_pj, int64_t h_pj_plane, int h_pj_kp) {
    AMS_REQUIRE(block_fused_supported(Cin, Cexp, Cout, stride, 1, residual), "block kernel: unsupported shape Cin=%d Cexp=%d Cout=%d s=%d", Cin, Cexp,
                Cout, stride);
    BlkArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.w_exp = w_exp; a.sc_e = sc_e; a.sh_e = sh_e; a.act_e = act_e; a.Cexp = Cexp;
    a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.w_pj = w_pj; a.sc_p = sc_p; a.sh_p = sh_p; a.act_p = act_p; a.Cout = Cout;
    a.residual = residual ? 1 : 0; a.y = y; a.vecs = vecs; a.timed = knobs().blk_timed;
    if (h_exp && h_pj) { a.he = h_exp; a.he_plane = h_exp_plane; a.hp = h_pj; a.hp_plane = h_pj_plane; a.kp_p = h_pj_kp; }      // fp16 form: expand and project
    else if (wparts && Cin > 16) { a.wparts = wparts; a.wplane = wplane; }      // X6 pays from K = 24 on (at K = 16 half of every bf16 MFMA is padding)
    same_pad(H, 3, stride, 1, &a.Ho, &a.pt);
    same_pad(W, 3, stride, 1, &a.Wo, &a.pl);
    // tile of output pixels per WAVE: the measured choice, or the next smaller one that fits LDS (blk_pick_tile)
    int th, tw;
    AMS_REQUIRE(blk_pick_tile(Cexp, Cout, stride, a.wparts != nullptr || a.he != nullptr, &th, &tw), "block kernel: no tile fits LDS for Cexp=%d", Cexp);
    if (knobs().blk_th > 0 && blk_lds_bytes(stride, knobs().blk_th, knobs().blk_tw, Cexp) <= 64 * 1024) {
        th = knobs().blk_th; tw = knobs().blk_tw;          // tuning knob AMS_BLK_TILE (tools/block_one.py); ignored when it does not fit
    }
#define BLK(S_, TH_, TW_) if (stride == S_ && th == TH_ && tw == TW_) return launch_blk_t<S_, TH_, TW_>(a, st);
    BLK(1, 8, 8) BLK(1, 4, 16) BLK(1, 4, 8) BLK(2, 4, 8) BLK(2, 2, 8) BLK(2, 4, 4)
#undef BLK
    set_error("block kernel: no %dx%d tile for stride %d", th, tw, stride);
    return AMS_E_INVALID;
}

// First block of the network in the same form: frames -> pad 127.5 -> x * ps - 1 -> stem 3x3 s2 (3 -> 32) + BN + ReLU6 -> depthwise 3x3 (32)
// + BN + ReLU6 -> project 1x1 (32 -> 16) + BN.  `vecs` = [13][32] table (stem BN | depthwise BN | depthwise taps) or null.
int launch_first_block_tiles(const void* frames, int dtype, int B, int H, int W, float pixel_scale, const float* w_stem, const float* sc_s,
                             const float* sh_s, int act_s, const float* w_dw, const float* sc_d, const float* sh_d, int act_d, const float* w_pj,
                             const float* sc_p, const float* sh_p, int act_p, float* y, hipStream_t st, const float* vecs) {
    AMS_REQUIRE(dtype == AMS_DT_U8 || dtype == AMS_DT_F32, "first_block: frames must be uint8 or float32");
    BlkArgs a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const float*>(frames); a.B = B; a.fH = H; a.fW = W; a.ps = pixel_scale; a.Cin = 27;
    a.w_exp = w_stem; a.sc_e = sc_s; a.sh_e = sh_s; a.act_e = act_s; a.Cexp = 32;
    a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.w_pj = w_pj; a.sc_p = sc_p; a.sh_p = sh_p; a.act_p = act_p; a.Cout = 16;
    a.residual = 0; a.y = y; a.vecs = vecs;
    same_pad(H + 1, 3, 2, 1, &a.H, &a.spt);           // the stem's output grid = the depthwise conv's input grid
    same_pad(W + 1, 3, 2, 1, &a.W, &a.spl);
    a.Ho = a.H; a.Wo = a.W; a.pt = 1; a.pl = 1;        // depthwise 3x3, stride 1, SAME
    if (dtype == AMS_DT_U8) return launch_blk_k<1, 2, 1, 8, 8, uint8_t>(a, st);
    return launch_blk_k<1, 2, 1, 8, 8, float>(a, st);
}

// tools/ only: read and clear the per-phase cycle sums of block_kernel (AMS_BLK_TIMED=1)
int blk_phase_cycles(unsigned long long* h) {
    static unsigned long long rows[1024][8], z[1024][8];
    if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_blk_cycles), sizeof(rows)) != hipSuccess) return AMS_E_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_blk_cycles), z, sizeof(z)) != hipSuccess) return AMS_E_HIP;
    for (int i = 0; i < 8; ++i) h[i] = 0;
    for (int r = 0; r < 1024; ++r)
        for (int i = 0; i < 8; ++i) h[i] += rows[r][i];
    return AMS_OK;
}

}  // namespace ams

// Fused expand (1x1, BN, ReLU6) -> depthwise 3x3 (BN, ReLU6) for the stride-1 blocks with 32 / 64 / 96 / 160 input channels
// (the two stride-8 blocks and the output-stride-16 section; rate 1 | 2), frozen inference, split-bf16 products: the STREAMING form.
//
// The tiled kernel of k_expand_dw.hip recomputes a 3x3 halo around every tile (x1.6 GEMM work at 8x8, x2.25 at rate 2)
// and runs the expand on the f32 matrix pipe; on these blocks it loses to the two separate kernels.  Here a block takes
// (frame, chunk of 16*NT expanded channels, sub-image, segment) and walks the segment's pixels in RASTER order, 64 per
// step.  The expanded values live in an LDS ring that spans two image rows + 64 pixels; the depthwise conv trails the
// GEMM by one row.  Nothing is recomputed along a row, only the 2 halo rows (and columns) of a segment.
//   * rate 2 = four independent rate-1 convolutions on the (row parity, column parity) sub-images: a pixel is a
//     contiguous Cin vector, so the strided gather is free, and the ring shrinks to two half-rows.
//   * the chunk's expand weights (NP bf16 parts, [part][k/8][n][8] so that ds_read_b128 is conflict-free without padding)
//     stay in LDS for the life of the block; depthwise taps and BN vectors of the thread's 4 channels stay in registers.
//   * E-step (per step): wave w owns pixels 16w .. 16w+15 of the step; operands of the NEXT step are requested before the
//     MFMAs of this one; products in the order of pw_gemm_bf16x3_l, so the result is bit-identical to the unfused pair.
//   * pixels outside the image (SAME zero padding of the depthwise conv, halo outside the sub-image) are written as 0: the
//     segment is laid out with one pad column on each side, so a tap never wraps into a neighbouring row and the
//     D-step needs no masks.
// The 6x-expanded tensor is never written; what reaches HBM is the depthwise output (the project GEMM's operand).
#include "pw_common.hpp"
#include "split_bf16.hpp"

namespace ams {

// the depthwise FMAs of the D-waves: scalar.  -DAMS_XDW_PK makes them v_pk_fma_f32 — measured neutral here (the D-waves share their SIMD
// with an E-wave that issues MFMAs all the time), unlike the whole-block kernel whose depthwise phase is its own phase
#ifdef AMS_XDW_PK
#define AMS_DW_FMA4(A_, V_, W_) fma4_pk(A_, V_, W_)
#else
#define AMS_DW_FMA4(A_, V_, W_) do { A_.x = fmaf(V_.x, W_.x, A_.x); A_.y = fmaf(V_.y, W_.y, A_.y); A_.z = fmaf(V_.z, W_.z, A_.z); A_.w = fmaf(V_.w, W_.w, A_.w); } while (0)
#endif

constexpr int xds_pitch(int nc) { return nc == 32 ? 48 : nc == 64 ? 80 : nc == 96 ? 112 : nc + 4; }

struct XdsArgs {
    const float* x;                  // [B, H, W, Cin]
    const unsigned short* xs;        // PRE: the same tensor as bf16 parts [part][B*H*W][Cin] (written by the producing GEMM), part p at xs + p * xs_plane
    int64_t xs_plane;
    int B, H, W, Cin;
    const unsigned short* wp;        // expand weights, bf16 parts [part][Cexp][Kp], part p at wp + p * plane
    int64_t plane;
    const float* wf;                 // F32: expand weights f32 [Cin][Cexp] (exact-f32 products for Cin <= 32)
    const float* sc_e; const float* sh_e;
    int act_e;
    int Cexp;
    const float* w_dw;               // [9][Cexp]
    const float* sc_d; const float* sh_d;
    int act_d;
    float* y;                        // [B, H, W, Cexp]
    int rate;
    int stride;                      // 1 | 2 (2: exact-f32 form only, rate 1); output [B, Ho, Wo, Cexp]
    int Ho, Wo, cy0, cx0;            // stride 2: output size; image row / column parity of the window centres (1 - pad before)
    int SH, SW;                      // rows / columns of a work item (sub-image coordinates; input pixels at stride 2)
    int Wp;                          // SW + 2: segment row pitch incl. one pad column on each side
    int T;                           // steps: ceil((SH + 2) * Wp / STEP)
    int ring;                        // ring size in pixels: a multiple of STEP, >= 2 * Wp + 2 + 2 * STEP
    int nsy, nsx;                    // segments per sub-image
    int chunks;                      // Cexp / (16 * NT)
    int items;                       // B * rate^2 * nsy * nsx work items per chunk
    int groups;                      // blocks per chunk; block g walks items g, g + groups, ...
    int y_fmt;                       // 0: y as f32; 1 (H16 only): y as fp16 pairs interleaved per 8 channels ("H2I", PwArgs::x_fmt) — same bytes
    int timed;                       // tools/ only (AMS_XWR_TIMED=1): per-role cycle sums into g_xds_cycles (ams_debug_phase_cycles(3, ..))
};

// Roles: waves [0, NWE) run the E-steps (operand loads, split, MFMAs, BN + ReLU6 into the ring), waves [NWE, NWE + NWD) the
// D-steps (depthwise from the ring, BN + ReLU6, stores).  D-step t - 1 runs beside E-step t, one barrier per step: the matrix
// pipe of a SIMD works for an E-wave while its vector pipe works for a D-wave, and neither role carries the other's registers.
// F32 (Cin <= 32: the early blocks): exact-f32 products on v_mfma_f32_16x16x4_f32, KS = number of 16-k chunks, k order of
// pw_gemm_f32_s / expand_dw_kernel (bit-identical to them); the contraction is so short that the f32 matrix pipe costs no more
// than six bf16 MFMAs plus the split.  Otherwise KS = 32-k stages of the split-bf16 product.
// H16: parts are the two fp16 parts of split_bf16.hpp (NP = 2): three MFMAs per 32 k with the cross terms in an accumulator of their own,
// products and order of pw_gemm_f16x3_l.
// tools/ only: [0] E-waves between the step barriers, [1] E-waves at the barrier, [2] / [3] the same for the D-waves, [6] E-wave steps, [7] D-wave steps
__device__ unsigned long long g_xds_cycles[1024][8];

template <int KS, int NT, int NP, int NWE, int NWD, bool PRE, bool F32, int S = 1, bool H16 = false>
__global__ __launch_bounds__(64 * (NWE + NWD)) void xdw_stream_kernel(XdsArgs a, unsigned nblocks) {
    static_assert(!H16 || (NP == 2 && !F32 && S == 1), "the fp16 form: two parts, stride 1");
    constexpr int NC = 16 * NT;                      // expanded channels per block
    constexpr int CG = NC / 4;                       // channel groups (float4) of the D-step
    constexpr int STEP = 16 * NWE;                   // pixels per step
    constexpr int PX = STEP * CG / (64 * NWD);       // consecutive centres per D-thread
    static_assert(PX * 64 * NWD == STEP * CG && PX >= 1, "D-step mapping");
    static_assert(!(F32 && PRE), "the exact-f32 form splits nothing");
    // ring row pitch in floats, chosen for the D-waves' tap reads (12 ds_read_b128 per thread and step against the E-waves' few stores):
    // under the real lane groups of ds_read_b128 (MI355X_MICROARCH.md) NC + 4 is 2- to 3-way conflicted, these pitches are conflict-free
    // for the reads and 2-way for the stores (brute-forced over the lane -> (pixel group, channel group) map of the D-step)
    constexpr int PITCH = xds_pitch(NC);
    constexpr int MIRROR = 4;                        // ring slots repeated after the end (a run of taps is <= 4 slots)
    constexpr int Kp = F32 ? KS * 16 : KS * 32;
    constexpr int WPF = NC + 4;                      // F32: pitch of the weight rows [k][n]
    constexpr int W_BYTES = F32 ? Kp * WPF * 4 : NP * Kp * NC * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* sW = reinterpret_cast<u32x4*>(smem);                      // [NP][KS][4][NC] 16-byte units (8 k each)
    float* sWf = reinterpret_cast<float*>(smem);                     // F32: [Kp][WPF]
    float* sAff = reinterpret_cast<float*>(smem + W_BYTES);          // sc_e, sh_e : 2 x NC
    float* ring = sAff + 2 * NC;                                     // [ring][PITCH]

    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int chunk = lb % a.chunks;
    const int group = lb / a.chunks;
    const int n0 = chunk * NC;                                       // the last chunk may be short (Cexp = 144: 16 of 32 channels)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the role branch is wave-uniform for the compiler too
    const int Wp = a.Wp, rate = a.rate, R = a.ring;

    // ---- chunk parameters -> LDS (once per block); channels past Cexp repeat the last one and are never stored -------------
    if constexpr (F32) {
        for (int e = tid; e < Kp * NC; e += 64 * (NWE + NWD)) {
            const int k = e / NC, n = e - k * NC;
            const int nn = n0 + n < a.Cexp ? n0 + n : a.Cexp - 1;
            sWf[k * WPF + n] = k < a.Cin ? a.wf[(int64_t)k * a.Cexp + nn] : 0.f;
        }
    } else {
        constexpr int UPR = Kp / 8;                                  // 16-byte units per weight row
        constexpr int NPIECE = NP * NC * UPR;
        for (int e = tid; e < NPIECE; e += 64 * (NWE + NWD)) {
            const int part = e / (NC * UPR), r = e - part * (NC * UPR), n = r / UPR, u = r - n * UPR;
            const int nn = n0 + n < a.Cexp ? n0 + n : a.Cexp - 1;
            const u32x4 v = *reinterpret_cast<const u32x4*>(a.wp + part * a.plane + (int64_t)nn * Kp + u * 8);
            sW[((part * KS + (u >> 2)) * 4 + (u & 3)) * NC + n] = v;
        }
    }
    if (tid < NC) {
        const int nn = n0 + tid < a.Cexp ? n0 + tid : a.Cexp - 1;
        sAff[tid] = a.sc_e[nn]; sAff[NC + tid] = a.sh_e[nn];
    }
    __syncthreads();
    const int qS = STEP / Wp, rS = STEP - qS * Wp;                   // both walkers advance by STEP pixels a step
    unsigned long long tc[2] = {0, 0}, tl_ = a.timed ? __builtin_amdgcn_s_memtime() : 0, nstep = 0;
    auto lap = [&](int slot) {
        if (a.timed) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tc[slot] += now - tl_;
            tl_ = now;
        }
    };
    auto flush = [&](int base) {
        if (a.timed && lane == 0) {
            unsigned long long* row = g_xds_cycles[(blockIdx.x * 8 + wave) & 1023];
            atomicAdd(&row[base], tc[0]);
            atomicAdd(&row[base + 1], tc[1]);
            atomicAdd(&row[base ? 7 : 6], nstep);
        }
    };

    if (wave < NWE) {
        // =================================== E-waves ===================================
        const int l15 = lane & 15, q = lane >> 4;
        const int ring_lane = (16 * wave + l15) * PITCH + 4 * q;     // this lane's slot within a STEP-aligned ring chunk
        int sbase = 0;                                               // (STEP * step counter) mod R, carried across items
        for (int item = group; item < a.items; item += a.groups) {
            int u1 = item;
            const int segx = u1 % a.nsx; u1 /= a.nsx;
            const int segy = u1 % a.nsy; u1 /= a.nsy;
            const int sub = u1 % (rate * rate);
            const int b = u1 / (rate * rate);
            const int sy = sub / rate, sx = sub - sy * rate;
            const int Hs = (a.H - sy + rate - 1) / rate, Ws = (a.W - sx + rate - 1) / rate;      // this sub-image
            const int i0 = segy * a.SH, j0 = segx * a.SW;
            const bool live = i0 < Hs && j0 < Ws;                    // a segment past the end of a smaller sub-image: barriers only
            // operand source of a pixel: f32 activations (split here) or, PRE, the bf16 parts the producing GEMM wrote beside them
            const int64_t frame0 = (int64_t)b * a.H * a.W * a.Cin + (F32 ? 0 : 8 * q);
            int e_row, e_col;
            { const int e0 = 16 * wave + l15; e_row = e0 / Wp; e_col = e0 - e_row * Wp; }
            float4 raw[(PRE || F32) ? 1 : KS][2];
            u32x4 rawp[PRE ? KS : 1][NP];
            float4 rawf[F32 ? KS : 1];                                // F32: k = 16c + 4q .. + 3 of the lane's pixel
            bool in_next = false;
            auto next_pixel = [&]() -> int64_t {                     // clamped element offset; `in_next` decides what is kept
                const int i = i0 - 1 + e_row, j = j0 - 1 + e_col;
                in_next = (i >= 0) & (i < Hs) & (j >= 0) & (j < Ws) & (e_row < a.SH + 2) & live;
                const int ic = i < 0 ? 0 : (i > Hs - 1 ? Hs - 1 : i), jc = j < 0 ? 0 : (j > Ws - 1 ? Ws - 1 : j);
                return frame0 + ((int64_t)(sy + rate * ic) * a.W + (sx + rate * jc)) * a.Cin;
            };
            auto load_stage = [&](int64_t off, int s) {
                if constexpr (F32) {
                    int ko = 16 * s + 4 * q;                          // k past Cin meets zero weight rows: any finite value will do
                    if (ko > a.Cin - 4) ko = a.Cin - 4;
                    rawf[s] = ld4(a.x + off + ko);
                } else if constexpr (PRE) {
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) rawp[s][pp] = *reinterpret_cast<const u32x4*>(a.xs + pp * a.xs_plane + off + 32 * s);
                } else {
                    raw[s][0] = ld4(a.x + off + 32 * s);
                    raw[s][1] = ld4(a.x + off + 32 * s + 4);
                }
            };
            {
                const int64_t p = next_pixel();
#pragma unroll
                for (int s = 0; s < KS; ++s) load_stage(p, s);
            }
            for (int t = 0; t < a.T; ++t) {
                const bool inside = in_next;
                f32x4 acc[NT], accx[H16 ? NT : 1];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < (H16 ? NT : 1); ++tt) accx[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                e_row += qS; e_col += rS;
                if (e_col >= Wp) { e_col -= Wp; ++e_row; }
                const int64_t pn = next_pixel();                      // pixel of step t + 1
                // Operands are split stage by stage (12 registers live instead of 12 * KS) and a stage's registers are refilled
                // with the next step's operands as soon as they are split: those loads have the rest of the step to land.
                // Products in the order of pw_gemm_bf16x3_l per accumulator; consecutive MFMAs go to different accumulators (see there:
                // chaining one accumulator's products measured 2-3 % slower in this kernel).
                if constexpr (F32) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const float4 av[1] = {rawf[s]};
                        load_stage(pn, s);
                        f32x4(&acc2)[1][NT] = *reinterpret_cast<f32x4(*)[1][NT]>(&acc);
                        pw_chunk<1, NT, WPF>(acc2, av, sWf + (16 * s + 4 * q) * WPF + l15);
                    }
                } else {
                bf16x8 x0, x1, x2;
    #pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        if constexpr (PRE) {
                            x0 = __builtin_bit_cast(bf16x8, rawp[s][0]);
                            if (NP >= 2) x1 = __builtin_bit_cast(bf16x8, rawp[s][NP >= 2 ? 1 : 0]);
                            if (NP == 3) x2 = __builtin_bit_cast(bf16x8, rawp[s][NP - 1]);
                        } else if constexpr (H16) {
                            f16x8 h, l;
                            split8_f16(raw[s][0], raw[s][1], h, l);
                            x0 = __builtin_bit_cast(bf16x8, h); x1 = __builtin_bit_cast(bf16x8, l);
                        } else {
                            if (NP == 3) split8(raw[s][0], raw[s][1], x0, x1, x2);
                            else if (NP == 2) split8(raw[s][0], raw[s][1], x0, x1);
                            else split8(raw[s][0], raw[s][1], x0);
                        }
                        load_stage(pn, s);
                        const u32x4* bw = sW + (s * 4 + q) * NC + l15;
                        bf16x8 q0[NT], q1[NT], q2[NT];
    #pragma unroll
                        for (int tt = 0; tt < NT; ++tt) {
                            q0[tt] = *reinterpret_cast<const bf16x8*>(bw + 16 * tt);
                            if (NP >= 2) q1[tt] = *reinterpret_cast<const bf16x8*>(bw + KS * 4 * NC + 16 * tt);
                            if (NP == 3) q2[tt] = *reinterpret_cast<const bf16x8*>(bw + 2 * KS * 4 * NC + 16 * tt);
                        }
                        if constexpr (H16) {                              // cross terms (wl xh, wh xl), then the main term
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q1[tt]), __builtin_bit_cast(f16x8, x0), accx[tt], 0, 0, 0);
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q0[tt]), __builtin_bit_cast(f16x8, x1), accx[tt], 0, 0, 0);
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q0[tt]), __builtin_bit_cast(f16x8, x0), acc[tt], 0, 0, 0);
                        } else {
                        if (NP == 3) {                                    // smallest terms first
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q2[tt], x0, acc[tt], 0, 0, 0);
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0[tt], x2, acc[tt], 0, 0, 0);
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1[tt], x1, acc[tt], 0, 0, 0);
                        }
                        if (NP >= 2) {
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1[tt], x0, acc[tt], 0, 0, 0);
    #pragma unroll
                            for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0[tt], x1, acc[tt], 0, 0, 0);
                        }
    #pragma unroll
                        for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0[tt], x0, acc[tt], 0, 0, 0);
                        }
                    }
                }
                {
                    float* dst = ring + sbase * PITCH + ring_lane;
                    const bool mirror = sbase == 0 && 16 * wave + l15 < MIRROR;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const float4 sc = ld4(sAff + 16 * tt + 4 * q), sh = ld4(sAff + NC + 16 * tt + 4 * q);
                        float4 v;
                        // one v_med3 per value: the activation's bounds, both 0 for positions outside the image (SAME padding of the
                        // depthwise conv, halo outside the sub-image)
                        const float lo = inside ? (a.act_e == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f) : 0.f;
                        const float hi = inside ? (a.act_e == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf()) : 0.f;
                        if constexpr (H16) acc[tt] = combine_f16(acc[tt], accx[H16 ? tt : 0]);
                        const float4 bn = muladd4_pk(make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]), sc, sh);
                        v.x = __builtin_amdgcn_fmed3f(bn.x, lo, hi); v.y = __builtin_amdgcn_fmed3f(bn.y, lo, hi);
                        v.z = __builtin_amdgcn_fmed3f(bn.z, lo, hi); v.w = __builtin_amdgcn_fmed3f(bn.w, lo, hi);
                        st4(dst + 16 * tt, v);
                        // the first MIRROR slots of the ring are repeated after its end: a D-thread's run of taps never wraps
                        if (mirror) st4(dst + R * PITCH + 16 * tt, v);
                    }
                }
                sbase += STEP;
                if (sbase == R) sbase = 0;
                lap(0);
                __syncthreads();
                lap(1);
                ++nstep;
            }
            __syncthreads();                                          // the D-waves finish the item (their step T - 1)
        }
        flush(0);
    } else {
        // =================================== D-waves ===================================
        const int dt = tid - 64 * NWE;
        const int cg = dt % CG, pt = dt / CG;
        const bool chan_ok = n0 + 4 * cg < a.Cexp;                   // short last chunk
        const int nch = chan_ok ? n0 + 4 * cg : 0;
        float4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = ld4(a.w_dw + (int64_t)k * a.Cexp + nch);
        const float4 dsc = ld4(a.sc_d + nch), dsh = ld4(a.sh_d + nch);
        // byte offset of what the thread stores within a pixel: its four f32 channels, or (y_fmt 1) one 16-byte half of its 8-channel group (k_xdw_wreg.hip)
        const bool odd_cg = (cg & 1) != 0;                          // = the lane's parity (CG is even)
        const unsigned ych = a.y_fmt ? (unsigned)(nch >> 3) * 32u + (odd_cg ? 16u : 0u) : (unsigned)nch * 4u;
        // ring position of this thread's first tap, (first centre - Wp - 1) mod R, carried across items like sbase
        int cb = (2 * R - 2 * Wp - 2 + pt * PX) % R;
        for (int item = group; item < a.items; item += a.groups) {
            int u1 = item;
            const int segx = u1 % a.nsx; u1 /= a.nsx;
            const int segy = u1 % a.nsy; u1 /= a.nsy;
            const int sub = u1 % (rate * rate);
            const int b = u1 / (rate * rate);
            const int sy = sub / rate, sx = sub - sy * rate;
            const int Hs = (a.H - sy + rate - 1) / rate, Ws = (a.W - sx + rate - 1) / rate;
            const int i0 = segy * a.SH, j0 = segx * a.SW;
            const bool live = i0 < Hs && j0 < Ws && chan_ok;
            // Results leave through buffer stores on a descriptor of this frame: a lane with nothing to store gets an offset
            // past the end, which the hardware drops.  No branch around the store, so hipcc counts it exactly.
            const int64_t oframe = S == 2 ? (int64_t)a.Ho * a.Wo * a.Cexp : (int64_t)a.H * a.W * a.Cexp;      // output elements per frame
            const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + b * oframe, 0, (int)(oframe * 4), 0x00020000);
            // valid output rows / columns of the item, and the byte offset of centre (row, col) = off0 + row * rowpitch + col * colpitch
            const int rmax = live ? (a.SH < Hs - i0 ? a.SH : Hs - i0) : 0, cmax = a.SW < Ws - j0 ? a.SW : Ws - j0;
            const int colpitch = rate * a.Cexp * 4, rowpitch = rate * a.W * a.Cexp * 4;
            const int off0 = ((sy + rate * (i0 - 1)) * a.W + sx + rate * (j0 - 1)) * a.Cexp * 4 + (int)ych;
            int d_row, d_col;                                        // first centre of this thread at step 0: -Wp - 1 + pt * PX
            { const int c0 = Wp - 1 + pt * PX; d_row = c0 / Wp; d_col = c0 - d_row * Wp; d_row -= 2; }
            __syncthreads();                                          // E-step 0
            for (int t = 0; t < a.T; ++t) {
                // D-step t: the STEP centres whose neighbourhood is complete after E-step t
                if constexpr (S == 2) {
                    // stride 2: the window centres sit on every other row and column of the input, so of a thread's pair of adjacent
                    // positions at most one is a centre, and whole steps fall on rows without any (wave-uniform skip)
#pragma unroll
                    for (int h = 0; h < PX; h += 2) {
                        int row = d_row, col = d_col + h;
#pragma unroll
                        for (int w = 0; w < (PX + 1) / 2; ++w)
                            if (col >= Wp) { col -= Wp; ++row; }
                        const int u = (j0 + col - 1 - a.cx0) & 1;     // 0: the first position of the pair is on a centre column
                        col += u;
                        if (col >= Wp) { col -= Wp; ++row; }
                        const int i = i0 + row - 1, j = j0 + col - 1;
                        const bool ok = ((unsigned)(row - 1) < (unsigned)rmax) & ((unsigned)(col - 1) < (unsigned)cmax) &
                                        ((((i - a.cy0) | (j - a.cx0)) & 1) == 0);
                        if (__builtin_amdgcn_ballot_w64(ok) != 0) {
                            float4 v[3][3];
#pragma unroll
                            for (int di = 0; di < 3; ++di) {
                                unsigned slot = (unsigned)(cb + di * Wp + h + u);
                                slot = slot < (unsigned)R ? slot : slot - (unsigned)R;
                                const float* rp = ring + __umul24(slot, PITCH) + 4 * cg;
#pragma unroll
                                for (int jj = 0; jj < 3; ++jj) v[di][jj] = ld4(rp + jj * PITCH);
                            }
                            float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                            for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                                for (int jj = 0; jj < 3; ++jj) {
                                    const float4 vv = v[ii][jj];
                                    const float4 w4 = wv[ii * 3 + jj];
                                    AMS_DW_FMA4(acc4, vv, w4);
                                }
                            float4 o;
                            o.x = apply_act(acc4.x * dsc.x + dsh.x, a.act_d); o.y = apply_act(acc4.y * dsc.y + dsh.y, a.act_d);
                            o.z = apply_act(acc4.z * dsc.z + dsh.z, a.act_d); o.w = apply_act(acc4.w * dsc.w + dsh.w, a.act_d);
                            const int oy = (i - a.cy0) >> 1, ox = (j - a.cx0) >> 1;
                            const unsigned off = ok ? (unsigned)((oy * a.Wo + ox) * a.Cexp * 4) + ych : 0xfffffff0u;
                            const u32x4 d = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                            __builtin_amdgcn_raw_buffer_store_b128(d, yrsrc, off, 0, 0);
                        }
                    }
                } else
#pragma unroll
                for (int h = 0; h < PX; h += 2) {                     // two centres at a time: 12 taps live
                    constexpr int PH = PX >= 2 ? 2 : 1;
                    float4 v[3][PH + 2];
#pragma unroll
                    for (int di = 0; di < 3; ++di) {
                        unsigned slot = (unsigned)(cb + di * Wp + h);         // first tap of the run; the run itself may pass the
                        slot = slot < (unsigned)R ? slot : slot - (unsigned)R; // end of the ring into the mirrored slots
                        const float* rp = ring + __umul24(slot, PITCH) + 4 * cg;
#pragma unroll
                        for (int jj = 0; jj < PH + 2; ++jj) v[di][jj] = ld4(rp + jj * PITCH);
                    }
#pragma unroll
                    for (int u = 0; u < PH; ++u) {
                        // scalar FMAs: v_pk_fma_f32 halves the instruction count but measured SLOWER here (+20 % on the kernel) —
                        // packed f32 shares a datapath with the MFMAs the E-waves keep in flight on the same SIMD
                        float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                const float4 vv = v[i][u + j];
                                const float4 w4 = wv[i * 3 + j];
                                AMS_DW_FMA4(acc4, vv, w4);
                            }
                        float4 o;
                        o.x = apply_act(acc4.x * dsc.x + dsh.x, a.act_d); o.y = apply_act(acc4.y * dsc.y + dsh.y, a.act_d);
                        o.z = apply_act(acc4.z * dsc.z + dsh.z, a.act_d); o.w = apply_act(acc4.w * dsc.w + dsh.w, a.act_d);
                        // centre (row, col) of the segment -> frame offset; (row - 1, col - 1) must lie inside the item's valid
                        // rows x columns (unsigned compare covers the pad row / column 0)
                        int row = d_row, col = d_col + h + u;
#pragma unroll
                        for (int w = 0; w < (PX + 1) / 2; ++w)            // PX may span several rows of a tiny segment (Wp >= 3)
                            if (col >= Wp) { col -= Wp; ++row; }
                        const bool ok = ((unsigned)(row - 1) < (unsigned)rmax) & ((unsigned)(col - 1) < (unsigned)cmax);
                        const unsigned off = ok ? (unsigned)(row * rowpitch + col * colpitch + off0) : 0xfffffff0u;
                        if (H16 && a.y_fmt) {                        // wave-uniform
                            // fp16 pairs, 8 channels per 32-byte group (16 bytes of hi | 16 bytes of lo).  Two neighbouring lanes hold the two
                            // halves of a group (channel groups cg = 2g, 2g + 1 of the same pixel): they swap — the even lane takes the odd one's
                            // hi, the odd lane the even one's lo (one quad_perm DPP move per dword) — and each stores ONE 16-byte piece instead of
                            // two 8-byte ones (the result stores are the largest single item of this kernel: tools/sweep_xwr_abl.sh)
                            unsigned h[2], l[2];
                            split4_f16(o, h, l);
                            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_cg ? h[0] : l[0]), 0xB1, 0xF, 0xF, false);
                            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_cg ? h[1] : l[1]), 0xB1, 0xF, 0xF, false);
                            const u32x4 d = {odd_cg ? r0 : h[0], odd_cg ? r1 : h[1], odd_cg ? l[0] : r0, odd_cg ? l[1] : r1};
                            __builtin_amdgcn_raw_buffer_store_b128(d, yrsrc, off, 0, 0);
                        } else {
                        const u32x4 d = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(d, yrsrc, off, 0, 0);
                        }
                    }
                }
                d_row += qS; d_col += rS;
                if (d_col >= Wp) { d_col -= Wp; ++d_row; }
                cb += STEP;
                if (cb >= R) cb -= R;
                lap(0);
                __syncthreads();                                      // E-step t + 1 (after the last step: the next item may start)
                lap(1);
                ++nstep;
            }
        }
        flush(2);
    }
}

// tools/ only: read and clear the per-role cycle sums of xdw_stream_kernel (AMS_XWR_TIMED=1)
int xds_phase_cycles(unsigned long long* h) {
    static unsigned long long rows[1024][8], z[1024][8];
    if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_xds_cycles), sizeof(rows)) != hipSuccess) return AMS_E_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_xds_cycles), z, sizeof(z)) != hipSuccess) return AMS_E_HIP;
    for (int i = 0; i < 8; ++i) h[i] = 0;
    for (int r = 0; r < 1024; ++r)
        for (int i = 0; i < 8; ++i) h[i] += rows[r][i];
    return AMS_OK;
}

// ---- host side -----------------------------------------------------------------------------------------------------
struct XdsPlan { int nt, nwe, nwd, SH, SW, nsy, nsx, ring, groups; size_t lds; };

// Kp = contraction length as staged (Cin rounded up to 32 for the split form, to 16 for the exact-f32 form: np = 0)
static size_t xds_lds(int Kp, int nt, int np, int ring) {
    const int NC = 16 * nt;
    const size_t w = np == 0 ? (size_t)Kp * (NC + 4) * 4 : (size_t)np * Kp * NC * 2;
    return w + 2 * NC * 4 + (size_t)(ring + 4) * xds_pitch(NC) * 4;      // + the mirrored slots
}
static int xds_ring(int SW, int step) { return (2 * (SW + 2) + 2 + 2 * step + step - 1) / step * step; }

// segment geometry: whole sub-image rows when the ring fits, column strips otherwise; row segments until there are enough
// work items to fill the chip (each costs two halo rows of GEMM work)
static bool xds_plan_try(int B, int H, int W, int Cin, int Cexp, int rate, int np, int nt, int nwe, int nwd, int nsy_force, int nsx_force,
                         int groups_force, XdsPlan* p) {
    const int Kp = np == 0 ? (Cin + 15) / 16 * 16 : Cin;
    const int Hs = (H + rate - 1) / rate, Ws = (W + rate - 1) / rate;
    const int step = 16 * nwe;
    const size_t budget = 160 * 1024 - 512;
    int nsx = nsx_force > 0 ? nsx_force : 1;
    int SW, ring;
    for (;; ++nsx) {
        SW = (Ws + nsx - 1) / nsx;
        ring = xds_ring(SW, step);
        // two blocks per CU when a narrower strip allows it
        if (xds_lds(Kp, nt, np, ring) <= budget / 2 || SW <= 24 || nsx_force > 0) break;
    }
    for (; xds_lds(Kp, nt, np, ring) > budget && SW > 1; ++nsx) {
        SW = (Ws + nsx) / (nsx + 1);
        ring = xds_ring(SW, step);
    }
    if (xds_lds(Kp, nt, np, ring) > budget || SW < 1) return false;
    nsx = (Ws + SW - 1) / SW;
    const int chunks = (Cexp + 16 * nt - 1) / (16 * nt);
    int nsy = 1;
    if (nsy_force > 0) nsy = nsy_force;
    else
        while ((int64_t)B * rate * rate * chunks * nsx * nsy < 768 && (Hs + nsy - 1) / nsy > 8) ++nsy;     // three blocks per CU (sweeps: bench_xds.py)
    p->nt = nt; p->nwe = nwe; p->nwd = nwd; p->nsx = nsx; p->SW = SW; p->nsy = nsy; p->SH = (Hs + nsy - 1) / nsy; p->ring = ring;
    p->lds = xds_lds(Kp, nt, np, ring);
    // blocks per chunk: enough to fill the chip a few times over, so that a block amortises its weight fill over several items
    const int64_t items = (int64_t)B * rate * rate * nsx * nsy;
    int64_t groups = groups_force > 0 ? groups_force : (2048 + chunks - 1) / chunks;
    if (groups > items) groups = items;
    if (groups < 1) groups = 1;
    p->groups = (int)groups;
    return true;
}

// segment geometry: whole sub-image rows when the ring fits, column strips otherwise; row segments until there are enough
// work items to fill the chip (each costs two halo rows of GEMM work).  AMS_XDS_FORCE = "tiles,row segments,column strips,
// E-waves,D-waves,blocks per chunk" overrides the choices (0 = automatic); a forced shape that does not fit LDS falls back.
static bool xds_plan(int B, int H, int W, int Cin, int Cexp, int rate, int np, XdsPlan* p) {
    // measured on MI355X at 32 frames (tools/bench_xds.py): 32-channel chunks and two blocks per CU for Cin 64 / 96; Cin 160 is
    // bound by the operand re-reads of its 30 chunk blocks (L1 / TA rate), 64-channel chunks halve them
    int nt = Cin >= 128 ? 4 : 2, nsy_force = 0, nsx_force = 0, nwe = 4, nwd = 4, groups_force = 0;
    if (knobs().xds_set) { const int* f = knobs().xds; nt = f[0]; nsy_force = f[1]; nsx_force = f[2]; nwe = f[3]; nwd = f[4]; groups_force = f[5]; }
    if (nt != 2 && nt != 4) nt = Cin >= 128 ? 4 : 2;
    if (np == 0) { nwe = 4; nwd = 4; }               // the exact-f32 form is built for 4 + 4 waves
    if (!((nwe == 4 && (nwd == 2 || nwd == 4)) || (nwe == 8 && nwd == 4))) { nwe = 4; nwd = 4; }
    if (xds_plan_try(B, H, W, Cin, Cexp, rate, np, nt, nwe, nwd, nsy_force, nsx_force, groups_force, p)) return true;
    if (xds_plan_try(B, H, W, Cin, Cexp, rate, np, 2, nwe, nwd, nsy_force, nsx_force, groups_force, p)) return true;
    return xds_plan_try(B, H, W, Cin, Cexp, rate, np, 2, 4, 4, nsy_force, 0, groups_force, p);
}

// Cin 16 / 24 / 32: exact-f32 products (the early blocks); 64 / 96 / 160: split-bf16 products (the stride-16 section)
bool expand_dw_stream_supported(int Cin, int Cexp, int stride, int rate) {
    if (rate != 1 && rate != 2) return false;
    if (stride == 2) { if (rate != 1 || Cin > 32) return false; }          // stride 2: the exact-f32 form only
    else if (stride != 1) return false;
    if (Cin != 16 && Cin != 24 && Cin != 32 && Cin != 64 && Cin != 96 && Cin != 160) return false;
    return Cexp % 16 == 0 && Cexp >= 32;
}

template <int KS, int NT, int NP, int NWE, int NWD, bool PRE, bool F32 = false, int S = 1, bool H16 = false>
static int launch_xds_p(XdsArgs a, const XdsPlan& p, hipStream_t st) {
    RUN_RC(func_allow_lds((const void*)xdw_stream_kernel<KS, NT, NP, NWE, NWD, PRE, F32, S, H16>, p.lds));
    const int64_t nblocks = (int64_t)a.groups * a.chunks;
    AMS_REQUIRE(nblocks > 0 && nblocks < 0x7fffffffLL, "expand_dw_stream: bad grid");
    static const std::string nm = "xdw_stream_kernel<" + std::to_string(KS) + ", " + std::to_string(NT) + ", " + std::to_string(NP) + ", " +
                                  std::to_string(NWE) + ", " + std::to_string(NWD) + (PRE ? ", true" : ", false") + (F32 ? ", true, " : ", false, ") + std::to_string(S) +
                                  (H16 ? ", true>" : ", false>");      // the symbol as rocprofv3 prints it
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((xdw_stream_kernel<KS, NT, NP, NWE, NWD, PRE, F32, S, H16>), dim3((unsigned)nblocks), dim3(64 * (NWE + NWD)), p.lds, st, a, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int KS, int NT, int NP, int NWE, int NWD>
static int launch_xds_k(const XdsArgs& a, const XdsPlan& p, hipStream_t st) {
    return a.xs ? launch_xds_p<KS, NT, NP, NWE, NWD, true>(a, p, st) : launch_xds_p<KS, NT, NP, NWE, NWD, false>(a, p, st);
}

template <int KS, int NT, int NP>
static int launch_xds_w(const XdsArgs& a, const XdsPlan& p, hipStream_t st) {
    if (p.nwe == 8) return launch_xds_k<KS, NT, NP, 8, 4>(a, p, st);
    if (p.nwd == 4) return launch_xds_k<KS, NT, NP, 4, 4>(a, p, st);
    return launch_xds_k<KS, NT, NP, 4, 2>(a, p, st);
}

// exact-f32 form: KC 16-k chunks, 4 E-waves + 4 D-waves
template <int KC>
static int launch_xds_f32(const XdsArgs& a, const XdsPlan& p, hipStream_t st) {
    if (a.stride == 2) return p.nt == 4 ? launch_xds_p<KC, 4, 3, 4, 4, false, true, 2>(a, p, st) : launch_xds_p<KC, 2, 3, 4, 4, false, true, 2>(a, p, st);
    if (p.nt == 4) return launch_xds_p<KC, 4, 3, 4, 4, false, true>(a, p, st);
    return launch_xds_p<KC, 2, 3, 4, 4, false, true>(a, p, st);
}

// fp16 form: 4 + 4 waves (what the plan picks unless forced)
template <int KS>
static int launch_xds_h16(const XdsArgs& a, const XdsPlan& p, hipStream_t st) {
    if (p.nt == 4) return a.xs ? launch_xds_p<KS, 4, 2, 4, 4, true, false, 1, true>(a, p, st) : launch_xds_p<KS, 4, 2, 4, 4, false, false, 1, true>(a, p, st);
    return a.xs ? launch_xds_p<KS, 2, 2, 4, 4, true, false, 1, true>(a, p, st) : launch_xds_p<KS, 2, 2, 4, 4, false, false, 1, true>(a, p, st);
}

template <int KS>
static int launch_xds_ks(const XdsArgs& a, const XdsPlan& p, int np, hipStream_t st) {
    if (p.nt == 4) return np == 3 ? launch_xds_w<KS, 4, 3>(a, p, st) : np == 1 ? launch_xds_w<KS, 4, 1>(a, p, st) : launch_xds_w<KS, 4, 2>(a, p, st);
    return np == 3 ? launch_xds_w<KS, 2, 3>(a, p, st) : np == 1 ? launch_xds_w<KS, 2, 1>(a, p, st) : launch_xds_w<KS, 2, 2>(a, p, st);
}

// w_f32: the expand weights [Cin][Cexp] for the exact-f32 form (Cin <= 32; w_parts / np / x_parts are then unused).
// x_parts (optional): x as bf16 parts [part][B*H*W][Cin], x_plane apart, exactly what split8 makes of x (the producing GEMM writes
// them, PwArgs::ysplit): the E-waves then load their operands ready-made instead of splitting them once per channel chunk.
// w_parts: the expand layer's bf16 panels [part][Cexp][Cin] (np = 2: hi, lo; np = 3: hi, mid, lo), part p at w_parts + p * plane
int launch_expand_dw_stream(const float* x, const uint16_t* x_parts, int64_t x_plane, int B, int H, int W, int Cin, const float* w_f32,
                            const uint16_t* w_parts, int64_t plane, int np,
                            const float* sc_e, const float* sh_e, int act_e, int Cexp, const float* w_dw, int stride, int rate, const float* sc_d,
                            const float* sh_d, int act_d, float* y, hipStream_t st, int y_fmt) {
    const bool f32 = Cin <= 32;
    const bool h16 = !f32 && np == AMS_NP_F16;       // two fp16 parts
    if (h16) np = 2;
    AMS_REQUIRE(y_fmt == 0 || (h16 && Cexp % 8 == 0 && stride == 1), "expand_dw_stream: the fp16-pair output needs the fp16 form and Cexp %% 8 == 0");
    AMS_REQUIRE(expand_dw_stream_supported(Cin, Cexp, stride, rate) && (f32 ? w_f32 != nullptr : (w_parts && np >= 1 && np <= 3)),
                "expand_dw_stream: unsupported shape Cin=%d Cexp=%d rate=%d", Cin, Cexp, rate);
    if (f32) { np = 0; x_parts = nullptr; }
    AMS_REQUIRE(B > 0 && H > 0 && W > 0, "expand_dw_stream: empty input");
    AMS_REQUIRE((int64_t)H * W * Cexp * 4 < 0x7fffffffLL, "expand_dw_stream: a frame of the output exceeds 2 GiB");
    XdsPlan p;
    AMS_REQUIRE(xds_plan(B, H, W, Cin, Cexp, rate, np, &p), "expand_dw_stream: no segment geometry fits LDS (W=%d rate=%d)", W, rate);
    XdsArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.xs = x_parts; a.xs_plane = x_plane; a.wf = w_f32; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.wp = w_parts; a.plane = plane; a.sc_e = sc_e; a.sh_e = sh_e; a.act_e = act_e;
    a.Cexp = Cexp; a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.y = y; a.rate = rate;
    a.stride = stride;
    if (stride == 2) {
        int pt, pl;
        same_pad(H, 3, 2, 1, &a.Ho, &pt);
        same_pad(W, 3, 2, 1, &a.Wo, &pl);
        a.cy0 = 1 - pt; a.cx0 = 1 - pl;
    }
    const int step = 16 * p.nwe;
    a.SH = p.SH; a.SW = p.SW; a.Wp = p.SW + 2; a.T = ((p.SH + 2) * a.Wp + step - 1) / step; a.ring = p.ring;
    a.nsy = p.nsy; a.nsx = p.nsx; a.chunks = (Cexp + 16 * p.nt - 1) / (16 * p.nt);
    a.items = B * rate * rate * p.nsy * p.nsx; a.groups = p.groups;
    a.y_fmt = y_fmt;
    a.timed = knobs().xwr_timed;
    if (h16) {
        AMS_REQUIRE(p.nwe == 4 && p.nwd == 4, "expand_dw_stream: the fp16 form runs with 4 + 4 waves");
        switch (Cin / 32) {
            case 2: return launch_xds_h16<2>(a, p, st);
            case 3: return launch_xds_h16<3>(a, p, st);
            default: return launch_xds_h16<5>(a, p, st);
        }
    }
    if (f32) return Cin <= 16 ? launch_xds_f32<1>(a, p, st) : launch_xds_f32<2>(a, p, st);
    switch (Cin / 32) {
        case 2: return launch_xds_ks<2>(a, p, np, st);
        case 3: return launch_xds_ks<3>(a, p, np, st);
        default: return launch_xds_ks<5>(a, p, np, st);
    }
}

}  // namespace ams

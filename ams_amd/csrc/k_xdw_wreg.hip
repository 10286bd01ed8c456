// Streaming expand + depthwise, WEIGHT-REGISTER form (the 160 -> 960 blocks; any Cin in 64 / 96 / 160 works).
//
// k_xdw_stream.hip keeps a 32- or 64-channel chunk of the expand weights in LDS and every chunk block reads the whole input:
// with 960 expanded channels that is 15..30 passes over the operand through the L1 / texture path, which is what bounds it
// there (rocprofv3: 11 % of the wave cycles issue instructions, the rest waits).  Here the roles of LDS and registers are
// swapped:
//   * an E-wave keeps ITS 32 channels of expand weights in registers for the life of the block (KS * 2 * NP fragments =
//     120 VGPRs at Cin = 160, three parts) — NWE waves, 32 * NWE channels per block;
//   * the operand of a 16-pixel step (bf16 parts written by the previous block's project GEMM, PwArgs::ysplit) is staged ONCE
//     per block in LDS by the E-waves (two 16-byte pieces per thread, double-buffered) and read by all of them: the input is
//     read 960 / (32 * NWE) times instead of 30;
//   * raster-order steps of 16 or 32 pixels (one or two MFMA row groups per E-wave: two give four accumulator chains), LDS ring of the expanded values, D-waves one step behind, one barrier per step — as in
//     k_xdw_stream.hip (same products in the same order, same depthwise order: bit-identical to it and to the unfused pair).
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

struct XwrArgs {
    const unsigned short* xs;        // operand as bf16 parts [part][B*H*W][Cin], part p at xs + p * xs_plane
    int64_t xs_plane;
    int B, H, W, Cin;
    const unsigned short* wp;        // expand weights, bf16 parts [part][Cexp][Cin], part p at wp + p * plane
    int64_t plane;
    const float* sc_e; const float* sh_e;
    int act_e;
    int Cexp;
    const float* w_dw;               // [9][Cexp]
    const float* sc_d; const float* sh_d;
    int act_d;
    float* y;                        // [B, H, W, Cexp]
    int rate;
    int SH, SW, Wp, T, ring;         // as in XdsArgs, with 16-pixel steps
    int nsy, nsx;
    int cgroups;                     // channel groups of 32 * NWE channels
    int items, groups;               // work items per channel group; blocks per channel group
    int y_fmt;                       // 0: y as f32; 1 (H16 only): y as fp16 pairs interleaved per 8 channels ("H2I", PwArgs::x_fmt) — same bytes
};

// tools/ only: [0] E-waves between barriers, [1] E-waves at the step barrier, [2] D-waves between barriers, [3] D-waves at the barrier,
// [4] D-waves from the barrier until their taps have landed, [5] from there until their FMAs are done, [2] the rest (epilogue + stores), [6] E-wave steps, [7] D-wave steps
__device__ unsigned long long g_xwr_cycles[1024][8];

// H16: the operand and weight parts are the two fp16 parts of split_bf16.hpp (hi | lo 2^11; NP = 2): three MFMAs per 32 k, the cross terms
// in an accumulator of their own, products and order of pw_gemm_f16x3_l (bit-identical to it followed by the depthwise kernel).
// ABL: measurement-only ablations (AMS_XWR_ABL, wrong results): 1 no operand loads in the step loop, 2 no MFMAs, 4 no depthwise arithmetic,
// 8 no result stores, 16 no ring stores
template <int KS, int NP, int NWE, int NWD, int NRG, bool H16 = false, int ABL = 0>
__global__ __launch_bounds__(64 * (NWE + NWD)) void xdw_wreg_kernel(XwrArgs a, unsigned nblocks) {
    static_assert(!H16 || NP == 2, "the fp16 form has two parts");
    constexpr int STEP = 16 * NRG;                   // pixels per step: NRG MFMA row groups per E-wave (2 * NRG accumulator chains)
    constexpr int NCB = 32 * NWE;                    // channels per block
    constexpr int CG = NCB / 4;                      // channel groups (float4) of the D-step
    constexpr int NDT = 64 * NWD;                    // D-threads
    constexpr int PX = STEP * CG / NDT;              // consecutive centres per D-thread, taken two at a time
    static_assert(PX * NDT == STEP * CG && NDT % CG == 0 && (PX == 1 || PX % 2 == 0), "D-step mapping");
    constexpr int PH = PX >= 2 ? 2 : 1;
    constexpr int PITCH = NCB + 4;
    constexpr int MIRROR = 4;                        // a run of PH + 2 taps may pass the end of the ring by PH + 1 slots
    constexpr int Kp = KS * 32;
    constexpr int PPP = NP * KS * 4;                 // 16-byte pieces of a pixel's operand: [part][s][q]
    constexpr int AUNITS = NRG * PPP * 16;           // 16-byte units of one operand tile: [rg][part][s][q][pixel]
    constexpr int TPP = 4 * NWE / NRG;               // loader threads per pixel
    constexpr int LPT = (PPP + TPP - 1) / TPP;       // pieces per loader thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* sA = reinterpret_cast<u32x4*>(smem);                      // [2][AUNITS]
    float* ring = reinterpret_cast<float*>(smem + 2 * AUNITS * 16);  // [ring + MIRROR][PITCH]

    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int cgi = lb % a.cgroups;
    const int group = lb / a.cgroups;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wp = a.Wp, rate = a.rate, R = a.ring;
    const int qS = STEP / Wp, rS = STEP - qS * Wp;
    constexpr bool TIMED = (ABL & 32) != 0;          // compile-time: the laps' state costs the untimed kernel 4 % when it is a run-time switch
    unsigned long long tc[4] = {0, 0, 0, 0}, tl_ = TIMED ? __builtin_amdgcn_s_memtime() : 0, nstep = 0;
    auto lap = [&](int slot) {
        if constexpr (TIMED) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tc[slot] += now - tl_;
            tl_ = now;
        }
    };
    auto flush = [&](int base) {
        if (TIMED && lane == 0) {
            unsigned long long* row = g_xwr_cycles[(blockIdx.x * 8 + wave) & 1023];
            atomicAdd(&row[base], tc[0]);
            atomicAdd(&row[base + 1], tc[1]);
            atomicAdd(&row[base ? 7 : 6], nstep);
            if (base) { atomicAdd(&row[4], tc[2]); atomicAdd(&row[5], tc[3]); }
        }
    };

    if (wave < NWE) {
        // =================================== E-waves ===================================
        const int l15 = lane & 15, q = lane >> 4;
        const int n0 = cgi * NCB + 32 * wave;                        // this wave's 32 channels
        const bool active = n0 < a.Cexp;                             // the last channel group may be short (wave-uniform)
        const int n0c = active ? n0 : 0;
        // ---- weights of the wave's channels -> registers (once per block) ----
        bf16x8 wq[NP][KS][2];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    wq[pp][s][tt] = *reinterpret_cast<const bf16x8*>(a.wp + pp * a.plane + (int64_t)(n0c + 16 * tt + l15) * Kp + 32 * s + 8 * q);
        float4 esc[2], esh[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { esc[tt] = ld4(a.sc_e + n0c + 16 * tt + 4 * q); esh[tt] = ld4(a.sh_e + n0c + 16 * tt + 4 * q); }
        const int ring_lane = l15 * PITCH + 32 * wave + 4 * q;
        const float lo_e = a.act_e == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_e = a.act_e == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
        // loader duty: pixel (lrg, l15) of a step is shared by TPP threads; thread lj of them takes pieces lj, lj + TPP, ...
        // (piece j = part * KS * 4 + s * 4 + qq).  Source offsets (elements, relative to the pixel) and LDS slots are fixed.
        const int lrg = NRG == 2 ? (q & 1) : 0;
        const int lj = NRG == 2 ? wave * 2 + (q >> 1) : wave * 4 + q;
        int64_t psrc[LPT];
        int pdst[LPT];
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            int j = lj + k * TPP;
            if (j > PPP - 1) j = PPP - 1;                            // surplus threads repeat the last piece
            const int part = j / (KS * 4), u = j - part * (KS * 4);
            psrc[k] = part * a.xs_plane + 8 * u;
            pdst[k] = (lrg * PPP + j) * 16 + l15;
        }
        int sbase = 0;
        int par = 0;                                                 // operand buffer of the current step
        for (int item = group; item < a.items; item += a.groups) {
            int u1 = item;
            const int segx = u1 % a.nsx; u1 /= a.nsx;
            const int segy = u1 % a.nsy; u1 /= a.nsy;
            const int sub = u1 % (rate * rate);
            const int b = u1 / (rate * rate);
            const int sy = sub / rate, sx = sub - sy * rate;
            const int Hs = (a.H - sy + rate - 1) / rate, Ws = (a.W - sx + rate - 1) / rate;
            const int i0 = segy * a.SH, j0 = segx * a.SW;
            const bool live = i0 < Hs && j0 < Ws;
            const unsigned short* xf = a.xs + (int64_t)b * a.H * a.W * a.Cin;
            // walkers: pixel l15 + 16 * rg of the step; both advance by STEP a step
            int e_row[NRG], e_col[NRG];
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) { const int e0 = l15 + 16 * rg; e_row[rg] = e0 / Wp; e_col[rg] = e0 - e_row[rg] * Wp; }
            bool in_cur[NRG];
            auto inside_of = [&](int rg) {
                const int i = i0 - 1 + e_row[rg], j = j0 - 1 + e_col[rg];
                return (bool)((i >= 0) & (i < Hs) & (j >= 0) & (j < Ws) & (e_row[rg] < a.SH + 2) & live);
            };
            auto loader_pixel = [&]() -> const unsigned short* {     // clamped address of this thread's loader pixel
                const int er = NRG == 2 ? (lrg ? e_row[NRG - 1] : e_row[0]) : e_row[0], ec = NRG == 2 ? (lrg ? e_col[NRG - 1] : e_col[0]) : e_col[0];
                const int i = i0 - 1 + er, j = j0 - 1 + ec;
                const int ic = i < 0 ? 0 : (i > Hs - 1 ? Hs - 1 : i), jc = j < 0 ? 0 : (j > Ws - 1 ? Ws - 1 : j);
                return xf + ((int64_t)(sy + rate * ic) * a.W + (sx + rate * jc)) * a.Cin;
            };
            u32x4 piece[LPT];
            auto load_pieces = [&](const unsigned short* px) {
#pragma unroll
                for (int k = 0; k < LPT; ++k) piece[k] = *reinterpret_cast<const u32x4*>(px + psrc[k]);
            };
            auto store_pieces = [&](int buf) {
#pragma unroll
                for (int k = 0; k < LPT; ++k) sA[buf * AUNITS + pdst[k]] = piece[k];
            };
            // prologue of the item: operand tile of step 0
            load_pieces(loader_pixel());
            store_pieces(par);
            __syncthreads();                                          // (A) tile 0 visible; pairs with the D-waves' first barrier
            for (int t = 0; t < a.T; ++t) {
#pragma unroll
                for (int rg = 0; rg < NRG; ++rg) {
                    in_cur[rg] = inside_of(rg);
                    e_row[rg] += qS; e_col[rg] += rS;
                    if (e_col[rg] >= Wp) { e_col[rg] -= Wp; ++e_row[rg]; }
                }
                if constexpr (!(ABL & 1)) load_pieces(loader_pixel());        // tile of step t + 1: lands during the MFMAs
                // The accumulators start at zero INSIDE the active branch (there the first MFMA of each takes the constant 0 as its C operand); a wave of
                // a short last channel group skips the products and the ring stores alike — nothing valid reads its ring columns (the D-threads of
                // those channels store nothing) — so no path needs 32 zeroed registers (they were 32 v_mov per step on every path)
                f32x4 acc[NRG][2], accx[H16 ? NRG : 1][2];
                auto zero_acc = [&]() {
#pragma unroll
                    for (int rg = 0; rg < NRG; ++rg) { acc[rg][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rg][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int rg = 0; rg < (H16 ? NRG : 1); ++rg) { accx[rg][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[rg][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                };
                if constexpr ((ABL & 2) != 0) zero_acc();
                else if (active) {
                    zero_acc();
                    const u32x4* ap = sA + par * AUNITS + q * 16 + l15;
                    // the operand fragments of k-step s + 1 are requested before the MFMAs of k-step s: with one E-wave per SIMD nothing
                    // else hides the LDS round trip (five exposed waits per step otherwise)
                    constexpr bool XA2 = NWD <= 4;        // (with two D-waves beside it on the SIMD the E-wave has no registers for the second set)
                    bf16x8 xa[XA2 ? 2 : 1][NP][NRG];
                    auto load_x = [&](int s, bf16x8 (&dst)[NP][NRG]) {
#pragma unroll
                        for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                            for (int pp = 0; pp < NP; ++pp) dst[pp][rg] = __builtin_bit_cast(bf16x8, ap[((rg * NP + pp) * KS + s) * 64]);
                    };
                    if constexpr (XA2) load_x(0, xa[0]);
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        if constexpr (XA2) { if (s + 1 < KS) load_x(s + 1, xa[(s + 1) & 1]); }
                        else load_x(s, xa[0]);
                        constexpr int XS = XA2 ? 1 : 0;
                        bf16x8 x0[NRG], x1[NRG], x2[NRG];
#pragma unroll
                        for (int rg = 0; rg < NRG; ++rg) {
                            x0[rg] = xa[s & XS][0][rg];
                            if (NP >= 2) x1[rg] = xa[s & XS][NP >= 2 ? 1 : 0][rg];
                            if (NP == 3) x2[rg] = xa[s & XS][NP - 1][rg];
                        }
                        // per accumulator the products of pw_gemm_bf16x3_l in its order (smallest terms first); consecutive MFMAs
                        // go to different accumulators
#define AMS_XWR_TERM(WP, XB)                                                                                         \
    _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg)                                                               \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                             \
            acc[rg][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[WP][s][tt], XB[rg], acc[rg][tt], 0, 0, 0);
#define AMS_XWR_TERM_H(ACC, WP, XB)                                                                                   \
    _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg)                                                               \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                             \
            ACC[H16 ? rg : 0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wq[WP][s][tt]), __builtin_bit_cast(f16x8, XB[rg]), ACC[H16 ? rg : 0][tt], 0, 0, 0);
                        if constexpr (H16) {
                            AMS_XWR_TERM_H(accx, 1, x0)
                            AMS_XWR_TERM_H(accx, 0, x1)
                            AMS_XWR_TERM_H(acc, 0, x0)
                        } else {
                        if (NP == 3) {
                            AMS_XWR_TERM(NP - 1, x0)
                            AMS_XWR_TERM(0, x2)
                            AMS_XWR_TERM(1, x1)
                        }
                        if (NP >= 2) {
                            AMS_XWR_TERM(NP >= 2 ? 1 : 0, x0)
                            AMS_XWR_TERM(0, x1)
                        }
                        AMS_XWR_TERM(0, x0)
                        }
#undef AMS_XWR_TERM
#undef AMS_XWR_TERM_H
                    }
                }
                if (active || (ABL & 2) != 0)
#pragma unroll
                for (int rg = 0; rg < NRG; ++rg) {
                    // one v_med3 per value: the activation's bounds, both 0 for a position outside the feature map (the depthwise
                    // conv's zero padding)
                    const float lo = in_cur[rg] ? lo_e : 0.f, hi = in_cur[rg] ? hi_e : 0.f;
                    float* dst = ring + (sbase + 16 * rg) * PITCH + ring_lane;
                    const bool mirror = sbase == 0 && rg == 0 && l15 < MIRROR;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        float4 v;
                        if constexpr (H16) acc[rg][tt] = combine_f16(acc[rg][tt], accx[H16 ? rg : 0][tt]);
                        const float4 bn = muladd4_pk(make_float4(acc[rg][tt][0], acc[rg][tt][1], acc[rg][tt][2], acc[rg][tt][3]), esc[tt], esh[tt]);
                        v.x = __builtin_amdgcn_fmed3f(bn.x, lo, hi);
                        v.y = __builtin_amdgcn_fmed3f(bn.y, lo, hi);
                        v.z = __builtin_amdgcn_fmed3f(bn.z, lo, hi);
                        v.w = __builtin_amdgcn_fmed3f(bn.w, lo, hi);
                        if constexpr (!(ABL & 16)) {
                        st4(dst + 16 * tt, v);
                        if (mirror) st4(dst + R * PITCH + 16 * tt, v);
                        }
                    }
                }
                store_pieces(par ^ 1);
                par ^= 1;
                sbase += STEP;
                if (sbase == R) sbase = 0;
                lap(0);
                __syncthreads();
                lap(1);
                if constexpr (TIMED) ++nstep;
            }
            __syncthreads();                                          // the D-waves finish the item
            par = 0;
        }
        flush(0);
    } else {
        // =================================== D-waves ===================================
        const int dt = tid - 64 * NWE;
        const int cg = dt % CG, pt = dt / CG;
        const int nch = cgi * NCB + 4 * cg;                          // first of this thread's 4 channels
        const bool chan_ok = nch < a.Cexp;                           // short last channel group
        const int nchc = chan_ok ? nch : 0;
        float4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = ld4(a.w_dw + (int64_t)k * a.Cexp + nchc);
        const float4 dsc = ld4(a.sc_d + nchc), dsh = ld4(a.sh_d + nchc);
        // byte offset of what the thread stores within a pixel: its four f32 channels, or (y_fmt 1) one 16-byte half of its 8-channel group —
        // 32-byte groups of 16 bytes hi | 16 bytes lo: the even lane of a pair stores the hi half, the odd lane the lo half (see the store)
        const bool odd_cg = (cg & 1) != 0;                          // = the lane's parity (CG is even)
        const unsigned ych = a.y_fmt ? (unsigned)(nchc >> 3) * 32u + (odd_cg ? 16u : 0u) : (unsigned)nchc * 4u;
        int cb = (2 * R - 2 * Wp - 2 + pt * PX) % R;                 // ring slot of the thread's first tap, carried across items
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
            const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)b * a.H * a.W * a.Cexp, 0,
                                                                                   (int)((int64_t)a.H * a.W * a.Cexp * 4), 0x00020000);
            const int rmax = live ? (a.SH < Hs - i0 ? a.SH : Hs - i0) : 0, cmax = a.SW < Ws - j0 ? a.SW : Ws - j0;
            const int colpitch = rate * a.Cexp * 4, rowpitch = rate * a.W * a.Cexp * 4;
            const int off0 = ((sy + rate * (i0 - 1)) * a.W + sx + rate * (j0 - 1)) * a.Cexp * 4 + (int)ych;
            int d_row, d_col;                                        // first centre of this thread at step 0: -Wp - 1 + pt * PX
            { const int c0 = Wp - 1 + pt * PX; d_row = c0 / Wp; d_col = c0 - d_row * Wp; d_row -= 2; }
            // byte offset of that centre's result, carried from centre to centre and from step to step by additions (the product form
            // row * rowpitch + col * colpitch cost two quarter-rate integer multiplies per centre)
            const int wrapdelta = rowpitch - Wp * colpitch;          // a centre that passes the end of a padded row
            const int stepdelta = qS * rowpitch + rS * colpitch;
            int d_off = d_row * rowpitch + d_col * colpitch + off0;
            __syncthreads();                                          // (A)
            __syncthreads();                                          // E-step 0
            // The taps of a step's centres: vt[hh][di][jj] = window row di of the pair of centres hh, all requested before the first is used (with
            // one D-wave per SIMD nothing else hides the LDS round trip of the second pair behind the first pair's arithmetic).
            // Measured and NOT kept (round 6): the top and middle rows of step t + 1 requested one step AHEAD (with Wp >= STEP they were written by
            // E-steps <= t; the D-waves then behind a bare s_barrier so that the requests stay in flight) — bit-identical, the wait behind the
            // barrier shrinks to the bottom row's 8 reads, and the kernel does not move (146 vs 147 us over four A/B rounds): the round-5 clocks'
            // "1.3 k cycles until the taps have landed" were not what bounds the step; the same on xdw_stream_kernel: 485 vs 452 us (slower).
            constexpr int NH = (PX + 1) / 2;
            float4 vt[NH][3][PH + 2];
            auto load_row = [&](int di) {
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
                    unsigned slot = (unsigned)(cb + di * Wp + 2 * hh);
                    slot = slot < (unsigned)R ? slot : slot - (unsigned)R;
                    slot = slot < (unsigned)R ? slot : slot - (unsigned)R;          // (cb + 2 Wp + PX may pass the end twice over on a tiny ring)
                    const float* rp = ring + __umul24(slot, PITCH) + 4 * cg;
#pragma unroll
                    for (int jj = 0; jj < PH + 2; ++jj) vt[hh][di][jj] = (ABL & 4) ? wv[di] : ld4(rp + jj * PITCH);
                }
            };
            // BN + activation + fp16 split + store of a step's centres (first centre at (row, col), byte offset off_c)
            auto epilogue = [&](const float4 (&accs)[NH][PH], int row, int col, int off_c) {
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
                    for (int u = 0; u < PH; ++u) {                    // centres in raster order: (hh, u) = centre 2 hh + u of the thread
                        const float4 a4 = accs[hh][u];
                        float4 o;
                        o.x = apply_act(a4.x * dsc.x + dsh.x, a.act_d); o.y = apply_act(a4.y * dsc.y + dsh.y, a.act_d);
                        o.z = apply_act(a4.z * dsc.z + dsh.z, a.act_d); o.w = apply_act(a4.w * dsc.w + dsh.w, a.act_d);
                        const bool ok = ((unsigned)(row - 1) < (unsigned)rmax) & ((unsigned)(col - 1) < (unsigned)cmax);
                        const unsigned off = (ok && !(ABL & 8)) ? (unsigned)off_c : 0xfffffff0u;
                        ++col; off_c += colpitch;                    // the next centre (one wrap at most per increment, whatever Wp)
                        if (col >= Wp) { col -= Wp; ++row; off_c += wrapdelta; }
                        if (H16 && a.y_fmt) {                        // wave-uniform
                            // fp16 pairs, 8 channels per 32-byte group (16 bytes of hi | 16 bytes of lo).  Two neighbouring lanes hold the two
                            // halves of a group (channel groups cg = 2g, 2g + 1 of the same pixel): they swap — the even lane takes the odd one's
                            // hi, the odd lane the even one's lo (one quad_perm DPP move per dword) — and each stores ONE 16-byte piece instead of
                            // two 8-byte ones (the result stores are the largest single item of this kernel: tools/sweep_xwr_abl.sh)
                            unsigned h2[2], l2[2];
                            split4_f16_mix(o, h2, l2);
                            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_cg ? h2[0] : l2[0]), 0xB1, 0xF, 0xF, false);
                            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd_cg ? h2[1] : l2[1]), 0xB1, 0xF, 0xF, false);
                            const u32x4 d = {odd_cg ? r0 : h2[0], odd_cg ? r1 : h2[1], odd_cg ? l2[0] : r0, odd_cg ? l2[1] : r1};
                            // non-temporal (aux 2): the 264 MB of a 32-frame launch pass through the caches once on their way to the project GEMM; with
                            // the hint they do not push that GEMM's weight panels and this kernel's operand out of L2 — the GEMM behind it 116 vs 123 us,
                            // this kernel 146 vs 149, the step +0.5 % over four leases.  (The same hint on the smaller results of xdw_stream_kernel: -1.8 %;
                            // on the GEMM's operand loads: -4 %.)
                            __builtin_amdgcn_raw_buffer_store_b128(d, yrsrc, off, 0, 2);
                        } else {
                        const u32x4 d = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(d, yrsrc, off, 0, 0);
                        }
                    }
                }
            };
            for (int t = 0; t < a.T; ++t) {
                load_row(0);
                load_row(1);
                load_row(2);
                if constexpr (TIMED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lap(2); }     // [4]: the taps have landed
                float4 acc4[NH][PH];
#pragma unroll
                for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                    for (int u = 0; u < PH; ++u) acc4[hh][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                // per centre the taps in the order (row, column) of dw3x3_fwd_kernel
#pragma unroll
                for (int i = 0; i < ((ABL & 4) ? 1 : 3); ++i) {
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                        for (int u = 0; u < PH; ++u)
#pragma unroll
                            for (int j = 0; j < ((ABL & 4) ? 1 : 3); ++j) {
                                const float4 vv = vt[hh][i][u + j];
                                const float4 w4 = wv[i * 3 + j];
                                AMS_DW_FMA4(acc4[hh][u], vv, w4);
                            }
                }
                const int row0 = d_row, col0 = d_col, off0c = d_off;     // the step's first centre; the next ones follow by increments
                d_row += qS; d_col += rS; d_off += stepdelta;
                if (d_col >= Wp) { d_col -= Wp; ++d_row; d_off += wrapdelta; }
                cb += STEP;
                if (cb >= R) cb -= R;
                if constexpr (TIMED) {                                // [5]: the FMAs are done
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                        for (int u = 0; u < PH; ++u) asm volatile("" : "+v"(acc4[hh][u].x), "+v"(acc4[hh][u].y), "+v"(acc4[hh][u].z), "+v"(acc4[hh][u].w));
                    lap(3);
                }
                epilogue(acc4, row0, col0, off0c);
                lap(0);
                __syncthreads();
                lap(1);
                if constexpr (TIMED) ++nstep;
            }
        }
        flush(2);
    }
}

// tools/ only: read and clear the per-role cycle sums of xdw_wreg_kernel (AMS_XWR_TIMED=1)
int xwr_phase_cycles(unsigned long long* h) {
    static unsigned long long rows[1024][8], z[1024][8];
    if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_xwr_cycles), sizeof(rows)) != hipSuccess) return AMS_E_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_xwr_cycles), z, sizeof(z)) != hipSuccess) return AMS_E_HIP;
    for (int i = 0; i < 8; ++i) h[i] = 0;
    for (int r = 0; r < 1024; ++r)
        for (int i = 0; i < 8; ++i) h[i] += rows[r][i];
    return AMS_OK;
}

// ---- host side -----------------------------------------------------------------------------------------------------
static size_t xwr_lds(int Kp, int np, int nwe, int nrg, int ring) {
    return (size_t)2 * nrg * np * (Kp / 8) * 16 * 16 + (size_t)(ring + 4) * (32 * nwe + 4) * 4;
}

template <int KS, int NP, int NWE, int NWD, int NRG, bool H16 = false, int ABL = 0>
static int launch_xwr_k(const XwrArgs& a, size_t lds, hipStream_t st) {
#ifdef AMS_MEASURE
    if constexpr (ABL == 0 && KS == 5 && NP == 2 && NWE == 4 && NWD == 4 && NRG == 2 && H16) {        // MEASUREMENT BUILD ONLY (libams_hip_measure.so, AMS_XWR_ABL=<bits>)
        switch (knobs().xwr_abl) {
            case 1: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 1>(a, lds, st);
            case 2: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 2>(a, lds, st);
            case 3: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 3>(a, lds, st);
            case 4: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 4>(a, lds, st);
            case 8: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 8>(a, lds, st);
            case 12: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 12>(a, lds, st);
            case 16: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 16>(a, lds, st);
            case 19: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 19>(a, lds, st);
            case 28: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 28>(a, lds, st);
            case 31: return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 31>(a, lds, st);
            default: break;
        }
        if (knobs().xwr_timed) return launch_xwr_k<KS, NP, NWE, NWD, NRG, H16, 32>(a, lds, st);      // AMS_XWR_TIMED=1: the kernel with its role clocks
    }
#endif
    RUN_RC(func_allow_lds((const void*)xdw_wreg_kernel<KS, NP, NWE, NWD, NRG, H16, ABL>, lds));
    const int64_t nblocks = (int64_t)a.groups * a.cgroups;
    AMS_REQUIRE(nblocks > 0 && nblocks < 0x7fffffffLL, "expand_dw_wreg: bad grid");
    static const std::string nm = "xdw_wreg_kernel<" + std::to_string(KS) + ", " + std::to_string(NP) + ", " + std::to_string(NWE) + ", " +
                                  std::to_string(NWD) + ", " + std::to_string(NRG) + (H16 ? ", true, " : ", false, ") + std::to_string(ABL) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((xdw_wreg_kernel<KS, NP, NWE, NWD, NRG, H16, ABL>), dim3((unsigned)nblocks), dim3(64 * (NWE + NWD)), lds, st, a, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int KS, int NP, bool H16 = false>
static int launch_xwr_w(const XwrArgs& a, int nwe, int nrg, size_t lds, hipStream_t st) {
    if (nwe == 8) return launch_xwr_k<KS, NP, 8, 4, 1, H16>(a, lds, st);
    // (eight D-waves beside the four E-waves — <KS, NP, 4, 8, 2>, the operand fragments single-buffered to fit 168 VGPRs: measured in round 5, the
    // D-waves then finish a step in 2.7 k cycles instead of 4.0 k but the E-waves need 5.6 k instead of 3.2 k: 2.96 vs 2.74 ms per 32-frame pass)
    if (nrg == 2) return launch_xwr_k<KS, NP, 4, 4, 2, H16>(a, lds, st);
    return launch_xwr_k<KS, NP, 4, 4, 1, H16>(a, lds, st);
}

// the weight-register form; x_parts is required (bf16 parts of the input, np of them).  AMS_XWR_FORCE = "E-waves (4|8),row
// segments,column strips,blocks per channel group,row groups per E-wave and step (1|2; 2 only with 4 E-waves)" (0 = automatic).
int launch_expand_dw_wreg(const uint16_t* x_parts, int64_t x_plane, int B, int H, int W, int Cin, const uint16_t* w_parts, int64_t plane,
                          int np, const float* sc_e, const float* sh_e, int act_e, int Cexp, const float* w_dw, int rate, const float* sc_d,
                          const float* sh_d, int act_d, float* y, hipStream_t st, int y_fmt) {
    const bool h16 = np == AMS_NP_F16;               // two fp16 parts (split_bf16.hpp)
    if (h16) np = 2;
    AMS_REQUIRE(expand_dw_stream_supported(Cin, Cexp, 1, rate) && Cin >= 64 && np >= 1 && np <= 3 && x_parts, "expand_dw_wreg: unsupported shape Cin=%d Cexp=%d rate=%d",
                Cin, Cexp, rate);
    AMS_REQUIRE(y_fmt == 0 || (h16 && Cexp % 8 == 0), "expand_dw_wreg: the fp16-pair output needs the fp16 form and Cexp %% 8 == 0");
    AMS_REQUIRE(B > 0 && H > 0 && W > 0, "expand_dw_wreg: empty input");
    AMS_REQUIRE((int64_t)H * W * Cexp * 4 < 0x7fffffffLL, "expand_dw_wreg: a frame of the output exceeds 2 GiB");
    int nwe = 4, nsy_force = 0, nsx_force = 0, groups_force = 0, nrg = 2;
    if (knobs().xwr_set) { const int* f = knobs().xwr; nwe = f[0]; nsy_force = f[1]; nsx_force = f[2]; groups_force = f[3]; nrg = f[4]; }
    if (nwe != 4 && nwe != 8) nwe = 4;
    if (nrg != 1 && nrg != 2) nrg = 2;
    if (nwe == 8) nrg = 1;
    const int step = 16 * nrg;
    const int Hs = (H + rate - 1) / rate, Ws = (W + rate - 1) / rate;
    const size_t budget = 160 * 1024 - 512;
    int nsx = nsx_force > 0 ? nsx_force : 1, SW, ring;
    for (;; ++nsx) {
        SW = (Ws + nsx - 1) / nsx;
        ring = (2 * (SW + 2) + 2 + 2 * step + step - 1) / step * step;
        if (xwr_lds(Cin, np, nwe, nrg, ring) <= budget || SW <= 1) break;
    }
    AMS_REQUIRE(xwr_lds(Cin, np, nwe, nrg, ring) <= budget, "expand_dw_wreg: no segment geometry fits LDS (W=%d rate=%d)", W, rate);
    nsx = (Ws + SW - 1) / SW;
    const int cgroups = (Cexp + 32 * nwe - 1) / (32 * nwe);
    int nsy = 1;
    if (nsy_force > 0) nsy = nsy_force;
    else
        while ((int64_t)B * rate * rate * cgroups * nsx * nsy < 512 && (Hs + nsy - 1) / nsy > 8) ++nsy;
    XwrArgs a;
    memset(&a, 0, sizeof(a));
    a.xs = x_parts; a.xs_plane = x_plane; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.wp = w_parts; a.plane = plane; a.sc_e = sc_e; a.sh_e = sh_e;
    a.act_e = act_e; a.Cexp = Cexp; a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.y = y; a.rate = rate;
    a.SH = (Hs + nsy - 1) / nsy; a.SW = SW; a.Wp = SW + 2; a.T = ((a.SH + 2) * a.Wp + step - 1) / step; a.ring = ring;
    a.nsy = nsy; a.nsx = nsx; a.cgroups = cgroups;
    a.items = B * rate * rate * nsy * nsx;
    a.y_fmt = y_fmt;
    int64_t groups = groups_force > 0 ? groups_force : (512 + cgroups - 1) / cgroups;      // one block per CU (LDS), twice over
    if (groups > a.items) groups = a.items;
    a.groups = (int)groups;
    const size_t lds = xwr_lds(Cin, np, nwe, nrg, ring);
    if (h16) {
        switch (Cin / 32) {
            case 2: return launch_xwr_w<2, 2, true>(a, nwe, nrg, lds, st);
            case 3: return launch_xwr_w<3, 2, true>(a, nwe, nrg, lds, st);
            default: return launch_xwr_w<5, 2, true>(a, nwe, nrg, lds, st);
        }
    }
    switch (Cin / 32) {
        case 2: return np == 3 ? launch_xwr_w<2, 3>(a, nwe, nrg, lds, st) : np == 1 ? launch_xwr_w<2, 1>(a, nwe, nrg, lds, st) : launch_xwr_w<2, 2>(a, nwe, nrg, lds, st);
        case 3: return np == 3 ? launch_xwr_w<3, 3>(a, nwe, nrg, lds, st) : np == 1 ? launch_xwr_w<3, 1>(a, nwe, nrg, lds, st) : launch_xwr_w<3, 2>(a, nwe, nrg, lds, st);
        default: return np == 3 ? launch_xwr_w<5, 3>(a, nwe, nrg, lds, st) : np == 1 ? launch_xwr_w<5, 1>(a, nwe, nrg, lds, st) : launch_xwr_w<5, 2>(a, nwe, nrg, lds, st);
    }
}

}  // namespace ams

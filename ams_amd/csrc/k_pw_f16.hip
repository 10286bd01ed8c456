// 1x1 convolutions of frozen inference on TWO fp16 parts per operand (AMS_MATMUL_SPLIT_F16): see split_bf16.hpp for the split and its
// error bound, k_pw_x3.hip for the three-part bf16 form this replaces on the forward path (6 MFMAs per 32 k, 6 bytes per stored value —
// here 3 MFMAs and 4 bytes).
//
//   x ~ xh + xl 2^-11,  w ~ wh + wl 2^-11      (xh, xl, wh, wl fp16; xl, wl scaled by 2^11 so that they are normal numbers)
//   x w ~ xh wh + 2^-11 (xh wl + xl wh)         acc += wh xh;  accx += wl xh;  accx += wh xl;   result = acc + accx 2^-11
//
// The cross terms have an accumulator of their own (the matrix pipe has no scaled accumulate, and a 2^-11-scaled copy of one part would
// fall into fp16's subnormals); the two meet in the epilogue with one fma per value.
//
// Operand forms (PwArgs::x_fmt):
//   0  f32 activations [M][ldx], split in registers stage by stage (3 VALU per value);
//   1  "H2I": the producer already left the activation as fp16 pairs, interleaved per 8 channels — 16 bytes of hi, then 16 bytes of lo,
//      row pitch 4 C bytes: the SAME bytes per value and the same 32-byte lane pieces as f32 (so the operand path of this kernel, which
//      is what bounds it, is unchanged), and no split work at all.  The depthwise result of the stride-16 blocks is written this way by
//      the streaming expand + depthwise kernels (its only reader is this GEMM); round 3 tried the same hand-over with three bf16 planes
//      and lost to the 6 bytes per value.
// Weight panels [part][N][Kp] fp16 (hi, lo 2^11), Kp = K rounded up to 32, split once per ams_student_freeze.
#include <type_traits>

#include "pw_common.hpp"
#include "split_bf16.hpp"

namespace ams {

__global__ void split_w_f16_kernel(const float* __restrict__ w, int64_t sk, int64_t sn, int K, int N, int Kp, unsigned short* __restrict__ hi,
                                   unsigned short* __restrict__ lo) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    const float v = k < K ? w[k * sk + n * sn] : 0.f;
    unsigned short h, l;
    split1_f16(v, h, l);
    hi[i] = h;
    lo[i] = l;
}

int launch_split_weights_f16(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* lo, hipStream_t st) {
    const int64_t n = (int64_t)N * Kp;
    hipLaunchKernelGGL(split_w_f16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w, sk, sn, K, N, Kp, hi, lo);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// f32 [M][C] -> H2I (tests, and the fallback when a producer that cannot write the form feeds a consumer that wants it); C % 8 == 0
__global__ void pack_h2i_kernel(const float* __restrict__ x, int64_t n8, u32x4* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const float4 u = ld4(x + 8 * i), v = ld4(x + 8 * i + 4);
    f16x8 h, l;
    split8_f16(u, v, h, l);
    out[2 * i] = __builtin_bit_cast(u32x4, h);
    out[2 * i + 1] = __builtin_bit_cast(u32x4, l);
}

int launch_pack_h2i(const float* x, int64_t M, int C, float* out, hipStream_t st) {
    AMS_REQUIRE(C % 8 == 0 && M > 0, "pack_h2i: C (%d) must be a multiple of 8", C);
    const int64_t n8 = M * (C / 8);
    hipLaunchKernelGGL(pack_h2i_kernel, dim3((unsigned)cdiv64(n8, 256)), dim3(256), 0, st, x, n8, reinterpret_cast<u32x4*>(out));
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

__device__ __forceinline__ f32x4 mma_f16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

constexpr int pwh_waves(int rm, int nt, int nw) { return nw >= 9 ? 3 : nw == 8 ? 2 : rm * nt > 12 ? 1 : rm * nt > 6 ? 2 : 3; }
constexpr int pwh_epw(int nw, int slab) { return nw * slab <= 60 * 1024 ? nw : nw % 5 == 0 && 5 * slab <= 60 * 1024 ? 5 : nw % 4 == 0 ? 4 : nw % 3 == 0 ? 3 : 1; }
// blocks of 9 waves and more take their LDS dynamically (one block per CU anyway: 160 KB are theirs) and every wave has its own epilogue slab: ONE
// epilogue round instead of NW / EPW rounds of EPW waves with a block barrier between them
constexpr bool pwh_dyn(int nw, int epi) { return nw >= 9 && epi != EPI_GENERIC; }

// Same frame as pw_gemm_bf16x3_l (64 RM x 16 NT tiles, four waves with RM row groups each, weight stages double-buffered in XOR-swizzled
// 64-byte LDS rows, operand ring of depth 2 in registers, all loads unconditional, half-height tail blocks) with NP = 2 and two
// accumulators per tile.
// NW waves per block (4 | 8 | 12), each with its own 16 RM rows: a block's weight stages are staged once for all of them — with three
// 4-wave blocks per CU every stage of the panel crosses the L2 -> CU fabric three times, and at K = 960 x N = 160 the panels re-staged
// by 1074 blocks are MORE bytes than the activations (330 MB against 264).  D = stages of operands in flight per lane.
// ABL: measurement-only ablations (tools/sweep_pwh_abl.sh; wrong results): 1 no activation loads in the loop, 2 no weight loads / LDS stores,
// 4 no barrier, 8 no MFMAs
// XF = 1 (fine-tune step, forward): PwArgs::x_mode 1 — BN + activation of the layer that wrote x, applied between the operand's load and its
// split with the arithmetic of bn_act_kernel (k_pw_x3.hip has the same for its three parts); PwArgs::red_mode 1 (BN statistics of the
// result in the epilogue) is honoured with the plain epilogue in 4-wave blocks.
template <int RM, int NT, int EPI, int XP, int NW = 4, int D = 2, int ABL = 0, int XF = 0>
__global__ __launch_bounds__(64 * NW, pwh_waves(RM, NT, NW)) void pw_gemm_f16x3_l(PwArgs a, const unsigned short* __restrict__ w0, int64_t plane, int Kp,
                                                                                  int n_tiles_n, unsigned nblocks, unsigned n_full) {
    constexpr int NP = 2, NTH = 64 * NW;
    constexpr int PITCH = 32;
    constexpr int ROWS = 16 * NT;
    constexpr int NPIECE = NP * ROWS * 4;
    constexpr int NREG = (NPIECE + NTH - 1) / NTH;
    // the epilogue's per-wave slabs share the weight stages' LDS; when NW of them would pass 60 KB the waves take turns, EPW at a time
    constexpr int SLAB = 16 * (16 * NT + 4) * 4;
    constexpr bool DYN = pwh_dyn(NW, EPI);
    constexpr int EPW = (EPI == EPI_GENERIC || DYN) ? NW : pwh_epw(NW, SLAB);
    static_assert(NW % EPW == 0, "epilogue rounds");
    static_assert(XF == 0 || (XP == 0 && EPI == EPI_PLAIN), "the operand transform runs on f32 operands with the plain epilogue");
    // (+ 4 x 16 NT floats behind the slabs for the vectors of a fused column reduction, PwArgs::red_mode)
    constexpr int W_BYTES = 2 * NP * ROWS * PITCH * 2, OUT_BYTES = EPI == EPI_GENERIC ? 16 : EPW * SLAB + (EPI == EPI_PLAIN && NW == 4 ? 4 * 16 * NT * 4 : 0);
    __shared__ __attribute__((aligned(16))) unsigned char smem_static[DYN ? 16 : (W_BYTES > OUT_BYTES ? W_BYTES : OUT_BYTES)];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    unsigned char* smem = DYN ? smem_dyn : smem_static;
    typedef unsigned short (*WStage)[NP][ROWS * PITCH];
    WStage sW = reinterpret_cast<WStage>(smem);
    float* sOutAll = reinterpret_cast<float*>(smem);
    __shared__ __attribute__((aligned(16))) float sSc[16 * NT], sSh[16 * NT];
    constexpr int XVS = XF != 0 ? 1024 : 4;                          // floats per operand-transform vector (K <= XVS)
    __shared__ __attribute__((aligned(16))) float sXv[XF == 1 ? 2 * XVS : 4];
    const bool half = blockIdx.x >= n_full;
    const unsigned lb = half ? xcd_remap(blockIdx.x - n_full, nblocks - n_full) : xcd_remap(blockIdx.x, n_full);
    const int tile_n = lb % n_tiles_n;
    const int64_t tile_m = lb / n_tiles_n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int n0 = tile_n * ROWS;
    const int nrg = half ? RM / 2 : RM;
    const int64_t m_base = half ? (int64_t)(n_full / n_tiles_n) * (16 * NW * RM) + tile_m * (8 * NW * RM) + wave * (8 * RM)
                                : tile_m * (16 * NW * RM) + wave * (16 * RM);
    const int K = a.K, n_stages = Kp / 32;
    const int n_iter = (n_stages + D - 1) / D * D;

    u32x4 wring[D][NREG];
    const unsigned short* wsrc[NREG];
    int wdst[NREG];
#pragma unroll
    for (int u = 0; u < NREG; ++u) {
        const int e = tid + u * NTH < NPIECE ? tid + u * NTH : NPIECE - 1;
        const int which = e / (ROWS * 4), r = e - which * (ROWS * 4), n = r >> 2, part = r & 3;
        int nn = n0 + n;
        if (nn > a.N - 1) nn = a.N - 1;
        wsrc[u] = w0 + which * plane + (int64_t)nn * Kp + part * 8;
        wdst[u] = which * (ROWS * PITCH) + n * PITCH + (part ^ (((n >> 3) & 1) << 1)) * 8;
    }
    auto load_stage = [&](int s, u32x4 (&wreg)[NREG]) {
        if (s > n_stages - 1) s = n_stages - 1;
#pragma unroll
        for (int u = 0; u < NREG; ++u) wreg[u] = *reinterpret_cast<const u32x4*>(wsrc[u] + s * 32);
    };
    auto store_stage = [&](int buf, const u32x4 (&wreg)[NREG]) {
        unsigned short* base = &sW[buf][0][0];
#pragma unroll
        for (int u = 0; u < NREG; ++u) *reinterpret_cast<u32x4*>(base + wdst[u]) = wreg[u];
    };
    // a row of the operand is K floats (f32) or K fp16 pairs (H2I): 4 K bytes either way, and the lane's 8 k of a stage are the 32 bytes
    // at float offset 32 s + 8 q in both — 2 x float4 (f32: k .. k+3 | k+4 .. k+7; H2I: hi of the 8 k | lo of the 8 k)
    const float* arow[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        int64_t m = m_base + (r < nrg ? r : nrg - 1) * 16 + l15;
        if (m > a.M - 1) m = a.M - 1;
        arow[r] = a.x + m * (int64_t)a.ldx;
    }
    float4 abuf[D][RM][2];
    // ABL & 16 (probe, wrong operands): the same bytes requested in FULL-LINE lane order — a request instruction covers 8 rows x 128 bytes (lane ->
    // row lane / 8, 16-byte piece lane % 8) instead of 16 rows x 64 bytes: what the texture path would see behind a lane-order hop
    const float* aline[RM][2];
    if constexpr ((ABL & 16) != 0) {
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                int64_t m = m_base + (r < nrg ? r : nrg - 1) * 16 + 8 * hf + (lane >> 3);
                if (m > a.M - 1) m = a.M - 1;
                aline[r][hf] = a.x + m * (int64_t)a.ldx + 4 * (lane & 7);
            }
    }
    auto load_a = [&](int s, float4 (&dst)[RM][2], auto Rc) {
        constexpr int R = decltype(Rc)::value;
        int koff = s * 32 + 8 * q;
        if (koff > K - 8) koff = K - 8;                 // k >= K repeats the last 8 k of the row: the weight panels are zero there
        if constexpr ((ABL & 16) != 0) {
            int ks = s * 32;
            if (ks > K - 32) ks = K - 32;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                dst[r][0] = ld4(aline[r][0] + ks);
                dst[r][1] = ld4(aline[r][1] + ks);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dst[r][0] = ld4(arow[r] + koff);
            dst[r][1] = ld4(arow[r] + koff + 4);
        }
    };
    typedef std::integral_constant<int, RM> RFull;
    typedef std::integral_constant<int, (RM >= 2 ? RM / 2 : RM)> RHalf;
    load_stage(0, wring[0]);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
        if (d > 0) load_stage(d, wring[d]);
        load_a(d, abuf[d], RFull{});
    }
    f32x4 acc[RM][NT], accx[RM][NT];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    pw_stage_affine<NT>(a, sSc, sSh, n0, tid, NTH);
    if constexpr (XF != 0) {
        for (int e = tid; e < XVS; e += NTH) {
            const bool ok = e < K;
            sXv[e] = ok ? a.x_v0[e] : 0.f;
            sXv[XVS + e] = ok ? a.x_v1[e] : 0.f;
        }
    }
    store_stage(0, wring[0]);
    __syncthreads();
    auto main_loop = [&](auto Rc) {
        constexpr int R = decltype(Rc)::value;
        for (int s0 = 0; s0 < n_iter; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int s = s0 + d;
                if constexpr (!(ABL & 2)) load_stage(s + D - 1, wring[(d + D - 1) % D]);
                if constexpr (!(ABL & 1)) load_a(s + D - 1, abuf[(d + D - 1) % D], Rc);
                if (s < n_stages) {
                    u32x4 xh[RM], xl[RM];
                    float4 u0, u1, v0, v1;
                    if constexpr (XF != 0) {
                        int koff = s * 32 + 8 * q;
                        if (koff > K - 8) koff = K - 8;
                        u0 = ld4(sXv + koff); u1 = ld4(sXv + koff + 4); v0 = ld4(sXv + XVS + koff); v1 = ld4(sXv + XVS + koff + 4);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if constexpr (XF != 0) {
                            float4 y0 = muladd4_pk(abuf[d][r][0], u0, v0), y1 = muladd4_pk(abuf[d][r][1], u1, v1);
                            y0 = make_float4(apply_act(y0.x, a.x_act), apply_act(y0.y, a.x_act), apply_act(y0.z, a.x_act), apply_act(y0.w, a.x_act));
                            y1 = make_float4(apply_act(y1.x, a.x_act), apply_act(y1.y, a.x_act), apply_act(y1.z, a.x_act), apply_act(y1.w, a.x_act));
                            f16x8 h, l;
                            split8_f16(y0, y1, h, l);
                            xh[r] = __builtin_bit_cast(u32x4, h);
                            xl[r] = __builtin_bit_cast(u32x4, l);
                        } else if constexpr (XP) {
                            xh[r] = __builtin_bit_cast(u32x4, abuf[d][r][0]);
                            xl[r] = __builtin_bit_cast(u32x4, abuf[d][r][1]);
                        } else {
                            f16x8 h, l;
                            split8_f16(abuf[d][r][0], abuf[d][r][1], h, l);
                            xh[r] = __builtin_bit_cast(u32x4, h);
                            xl[r] = __builtin_bit_cast(u32x4, l);
                        }
                    }
                    const unsigned short* bw = &sW[s & 1][0][l15 * PITCH + 8 * (q ^ ((l15 >> 3) << 1))];      // LDS stages alternate with s (D may be odd)
                    // ABL & 32 (probe): ONE accumulator per tile (the cross terms join the main sum: other rounding, same instruction mix) and five tiles' fragments in flight
                    constexpr int TG = (ABL & 32) ? 5 : (NT >= 2 && RM <= 2) ? 2 : 1;
#pragma unroll
                    for (int t0 = 0; t0 < NT; t0 += TG) {
                        u32x4 qh[TG], ql[TG];
#pragma unroll
                        for (int g = 0; g < TG; ++g) {
                            if (t0 + g >= NT) continue;
                            qh[g] = *reinterpret_cast<const u32x4*>(bw + (t0 + g) * 16 * PITCH);
                            ql[g] = *reinterpret_cast<const u32x4*>(bw + (ROWS * PITCH) + (t0 + g) * 16 * PITCH);
                        }
                        // fixed order per accumulator pair: cross terms (wl xh, then wh xl), then the main term
#pragma unroll
                        for (int g = 0; g < TG; ++g)
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                if (t0 + g < NT && !(ABL & 8)) { if constexpr ((ABL & 32) != 0) acc[r][t0 + g] = mma_f16(ql[g], xh[r], acc[r][t0 + g]); else accx[r][t0 + g] = mma_f16(ql[g], xh[r], accx[r][t0 + g]); }
#pragma unroll
                        for (int g = 0; g < TG; ++g)
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                if (t0 + g < NT && !(ABL & 8)) { if constexpr ((ABL & 32) != 0) acc[r][t0 + g] = mma_f16(qh[g], xl[r], acc[r][t0 + g]); else accx[r][t0 + g] = mma_f16(qh[g], xl[r], accx[r][t0 + g]); }
#pragma unroll
                        for (int g = 0; g < TG; ++g)
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                if (t0 + g < NT && !(ABL & 8)) acc[r][t0 + g] = mma_f16(qh[g], xh[r], acc[r][t0 + g]);
                        if constexpr ((ABL & 8) != 0) {      // keep the fragment reads alive
#pragma unroll
                            for (int g = 0; g < TG; ++g) if (t0 + g < NT) { acc[0][t0 + g][0] += __uint_as_float(qh[g][0] ^ ql[g][1]); acc[0][t0 + g][1] += __uint_as_float(xh[0][0] ^ xl[RM - 1][1]); }
                        }
                    }
                }
                if constexpr (!(ABL & 2)) store_stage((s + 1) & 1, wring[(d + 1) % D]);
                if constexpr (!(ABL & 4)) __syncthreads();
            }
        }
    };
    if (RM >= 2 && half) main_loop(RHalf{});
    else main_loop(RFull{});
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) if constexpr (!(ABL & 32)) acc[r][t] = combine_f16(acc[r][t], accx[r][t]);
    const bool red = EPI == EPI_PLAIN && NW == 4 && a.red_mode != 0;    // block-uniform (the launcher clears red_mode where it cannot be fused)
    float4 rs1[(EPI == EPI_PLAIN && NW == 4) ? NT : 1], rs2[(EPI == EPI_PLAIN && NW == 4) ? NT : 1];
    if constexpr (EPI == EPI_PLAIN && NW == 4) {
        if (red) {
            // the stage loop ended with a barrier: the weight stages are dead, their LDS holds the reduction's vectors and, later, the waves' sums
            float* sRedVec = reinterpret_cast<float*>(smem) + 4 * 16 * (16 * NT + 4);
            pw_red_stage<NT>(a, sRedVec, n0, tid, NTH);
#pragma unroll
            for (int t = 0; t < NT; ++t) { rs1[t] = make_float4(0.f, 0.f, 0.f, 0.f); rs2[t] = rs1[t]; }
            __syncthreads();
            pw_red_rowgroups<RM, NT>(a, acc, m_base, n0, l15, q, sRedVec, rs1, rs2, nrg);
        }
    }
    if (EPI == EPI_GENERIC) pw_epilogue<RM, NT>(a, acc, m_base, n0, l15, q, sSc, sSh, nrg);
    else if constexpr (EPW == NW) pw_epilogue_t<RM, NT, EPI, true>(a, acc, m_base, n0, lane, sSc, sSh, sOutAll + wave * (16 * (16 * NT + 4)), nrg);
    else {
        for (int round = 0; round < NW / EPW; ++round) {            // block-uniform
            if (wave / EPW == round) pw_epilogue_t<RM, NT, EPI, true>(a, acc, m_base, n0, lane, sSc, sSh, sOutAll + (wave % EPW) * (16 * (16 * NT + 4)), nrg);
            __syncthreads();
        }
    }
    if constexpr (EPI == EPI_PLAIN && NW == 4) {
        if (red) {
            __syncthreads();                                        // the epilogue slabs are consumed
            const int64_t row = half ? (int64_t)(n_full / n_tiles_n) + tile_m : tile_m;
            pw_red_finish<NT>(a, rs1, rs2, lane, wave, 4, reinterpret_cast<float*>(smem), row, n0, tid, NTH);
        }
    }
}

template <int RM, int NT, int EPI, int XP, int NW = 4, int D = 2, int ABL = 0, int XF = 0>
static int launch_pw_f16_d(const PwArgs& a, const uint16_t* w, int64_t plane, int Kp, hipStream_t st) {
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    int per_cu = 1, cus = 256;
    // dynamic LDS of the wide blocks: max(weight stages, NW epilogue slabs) — the kernel's own W_BYTES / OUT_BYTES
    constexpr size_t kSlab = 16 * (16 * NT + 4) * 4, kW = (size_t)2 * 2 * 16 * NT * 32 * 2;
    const size_t dyn = pwh_dyn(NW, EPI) ? (NW * kSlab > kW ? NW * kSlab : kW) : 0;
    if (dyn) RUN_RC(func_allow_lds((const void*)pw_gemm_f16x3_l<RM, NT, EPI, XP, NW, D, ABL, XF>, dyn));
    RUN_RC(func_blocks_per_cu((const void*)pw_gemm_f16x3_l<RM, NT, EPI, XP, NW, D, ABL, XF>, 64 * NW, dyn, &per_cu));
    RUN_RC(device_cus(&cus));
    int64_t half_strips = 0;
    const int64_t full_strips = pw_plan_tail(a.M, RM, n_tiles_n, per_cu * cus, &half_strips, 16 * NW * RM);
    const int64_t n_full = full_strips * n_tiles_n;
    const int64_t nblocks = n_full + half_strips * n_tiles_n;
    PwArgs b = a;
    if (b.red_mode) {                               // partial rows: one per row strip of blocks
        if (EPI == EPI_PLAIN && NW == 4 && b.red_mode == 1 && pw_red_ok(b, nblocks / n_tiles_n)) { if (b.red_rows_out) *b.red_rows_out = (int)(nblocks / n_tiles_n); }
        else { b.red_mode = 0; if (b.red_rows_out) *b.red_rows_out = 0; }
    }
    static const std::string nm = "pw_gemm_f16x3_l<" + std::to_string(RM) + ", " + std::to_string(NT) + ", " + std::to_string(EPI) + ", " + std::to_string(XP) + ", " +
                                  std::to_string(NW) + ", " + std::to_string(D) + ", " + std::to_string(ABL) + ", " + std::to_string(XF) + ">";      // as rocprofv3 prints it
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_gemm_f16x3_l<RM, NT, EPI, XP, NW, D, ABL, XF>), dim3((unsigned)nblocks), dim3(64 * NW), dyn, st, b, w, plane, Kp, n_tiles_n,
                       (unsigned)nblocks, (unsigned)n_full);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int RM, int NT>
static int launch_pw_f16(const PwArgs& a, const uint16_t* w, int64_t plane, int Kp, hipStream_t st) {
    const int epi = pw_pick_epi(a);
    if (a.x_mode == 1) return launch_pw_f16_d<RM, NT, EPI_PLAIN, 0, 4, 2, 0, 1>(a, w, plane, Kp, st);      // (pointwise_f16_applies: plain epilogue only)
#ifdef AMS_MEASURE
    if constexpr (RM == 2 && NT == 5) {                    // MEASUREMENT BUILD ONLY (libams_hip_measure.so, AMS_PWH_ABL=<bits>, wrong results): what the stage loop is made of
        const int abl = knobs().pwh_abl;
        if (abl && a.x_fmt == 1) {
#define PWH_A(A_) if (abl == A_) return launch_pw_f16_d<RM, NT, EPI_PLAIN, 1, 4, 2, A_>(a, w, plane, Kp, st);
            PWH_A(1) PWH_A(2) PWH_A(3) PWH_A(4) PWH_A(6) PWH_A(7) PWH_A(8) PWH_A(9) PWH_A(10) PWH_A(11) PWH_A(15)
#undef PWH_A
        }
    }
#endif
    if constexpr (NT == 10 && RM <= 2) {                   // full-width 160-column tiles (the operand crosses L2 -> CU once), 8- / 12-wave blocks
        int nw = knobs().pwh_nw, dd = knobs().pwh_d;
        // what launch_pointwise_split_f16 picks the (1, 10) tile for.  160 columns (one column tile): 10-wave blocks — 160 rows a block: 429 blocks at
        // 68640 rows, 215 at the 34320 of a two-part plan, where 12-wave blocks are 358 / 179: fuller rounds on 256 CUs — 110 vs 117 us at 960 -> 160,
        // 64 vs 67 at 576 -> 160, 50.5 vs 53 at 34320 rows (round 6, AMS_PWH_VARIANT sweep); 320 columns keep 12 waves (182 vs 201 us)
        if (!knobs().pwh_set && RM == 1) { nw = a.N <= 160 ? 10 : 12; dd = 3; }
#ifdef AMS_MEASURE
        if constexpr (RM == 1) {                               // MEASUREMENT BUILD ONLY: the 12-wave form's stage loop taken apart (AMS_PWH_ABL, wrong results)
            const int abl = knobs().pwh_abl;
            if (abl && a.x_fmt == 1 && nw == 12 && dd == 3) {
#define PWH_A(A_) if (abl == A_) return launch_pw_f16_d<RM, NT, EPI_PLAIN, 1, 12, 3, A_>(a, w, plane, Kp, st);
                PWH_A(1) PWH_A(2) PWH_A(3) PWH_A(8) PWH_A(9) PWH_A(10) PWH_A(11) PWH_A(16) PWH_A(18) PWH_A(24) PWH_A(26) PWH_A(32) PWH_A(35) PWH_A(48) PWH_A(15) PWH_A(4)
#undef PWH_A
            }
        }
#endif
        if (a.x_fmt == 1 && (epi == EPI_PLAIN || epi == EPI_RES) && (nw > 4 || dd > 2)) {
#define PWH_V(NW_, D_) if (nw == NW_ && dd == D_) return epi == EPI_PLAIN ? launch_pw_f16_d<RM, NT, EPI_PLAIN, 1, NW_, D_>(a, w, plane, Kp, st) : launch_pw_f16_d<RM, NT, EPI_RES, 1, NW_, D_>(a, w, plane, Kp, st);
            PWH_V(8, 2) PWH_V(12, 2) PWH_V(8, 3) PWH_V(12, 3) PWH_V(10, 3)
#undef PWH_V
        }
    }
    if constexpr (RM == 2 && (NT == 5 || NT == 4)) {       // EXPERIMENT (AMS_PWH_VARIANT=<waves>,<depth>): wider blocks / deeper operand rings
        const int nw = knobs().pwh_nw, dd = knobs().pwh_d;
        if (a.x_fmt == 1 && (epi == EPI_PLAIN || epi == EPI_RES) && (nw > 4 || dd > 2)) {
#define PWH_V(NW_, D_) if (nw == NW_ && dd == D_) return epi == EPI_PLAIN ? launch_pw_f16_d<RM, NT, EPI_PLAIN, 1, NW_, D_>(a, w, plane, Kp, st) : launch_pw_f16_d<RM, NT, EPI_RES, 1, NW_, D_>(a, w, plane, Kp, st);
            PWH_V(4, 3) PWH_V(4, 4) PWH_V(8, 2) PWH_V(8, 3) PWH_V(12, 2) PWH_V(12, 3)
#undef PWH_V
        }
    }
    if (a.x_fmt == 1) {
        switch (epi) {
            case EPI_PLAIN: return launch_pw_f16_d<RM, NT, EPI_PLAIN, 1>(a, w, plane, Kp, st);
            case EPI_RES: return launch_pw_f16_d<RM, NT, EPI_RES, 1>(a, w, plane, Kp, st);
            case EPI_BIAS: return launch_pw_f16_d<RM, NT, EPI_BIAS, 1>(a, w, plane, Kp, st);
            default: return launch_pw_f16_d<RM, NT, EPI_GENERIC, 1>(a, w, plane, Kp, st);
        }
    }
    switch (epi) {
        case EPI_PLAIN: return launch_pw_f16_d<RM, NT, EPI_PLAIN, 0>(a, w, plane, Kp, st);
        case EPI_RES: return launch_pw_f16_d<RM, NT, EPI_RES, 0>(a, w, plane, Kp, st);
        case EPI_BIAS: return launch_pw_f16_d<RM, NT, EPI_BIAS, 0>(a, w, plane, Kp, st);
        default: return launch_pw_f16_d<RM, NT, EPI_GENERIC, 0>(a, w, plane, Kp, st);
    }
}

// the two-fp16-part product can take this problem (frozen inference: no fused reduction, no operand transform)
bool pointwise_f16_applies(const PwArgs& a) {
    if (!(a.M > 0 && a.K >= 32 && a.K % 8 == 0 && a.ldx % 4 == 0 && a.Kw == a.K && (a.x_fmt == 0 || a.ldx == a.K))) return false;
    if (a.red_mode > 1) return false;               // the backward's fused reduction stays with the three-part bf16 kernel (gradients leave fp16's range)
    if (a.x_mode == 0) return true;
    return a.x_mode == 1 && a.x_fmt == 0 && a.x_v0 && a.x_v1 && a.K <= 1024 && pw_pick_epi(a) == EPI_PLAIN;
}

// y = epilogue(x @ w), w as fp16 panels [part][N][Kp] (hi at whi, lo 2^11 at whi + plane)
int launch_pointwise_split_f16(const PwArgs& a, const uint16_t* whi, int64_t plane, int Kp, hipStream_t st) {
    if (a.red_rows_out) *a.red_rows_out = 0;               // set by the launch that fuses the column reduction (PwArgs::red_mode 1)
    AMS_REQUIRE(pointwise_f16_applies(a) && Kp % 32 == 0 && Kp >= a.K, "pointwise_split_f16: bad problem (M %lld K %d ldx %d)", (long long)a.M, a.K, a.ldx);
    // One to four frames per pass: a launch is one block's chain of K / 32 stages per tile (~0.33 us a stage).  Measured and NOT kept (round 5):
    // split-K inside a block — four waves on one 16-row x 32-column tile, every fourth stage each, weight fragments straight from the panels,
    // one LDS reduction in a fixed order — 0.530 vs 0.431 ms per one-frame call (675 blocks x 123 KB of panel reads instead of 170 x 123 KB, each
    // a dependent L2 round trip); three or four operand stages in flight in the tiled kernel 0.427 / 0.451 vs 0.432 ms
    int rm, nt;
    pw_pick_tile(a.M, a.N, &rm, &nt);
    if (rm == 2) {
        if (a.N == 96 || a.N == 960) nt = 6;
        else if (a.N == 320) nt = 5;
    }
    if (a.M <= 2400 && rm == 1 && a.N % 32 == 0 && a.N >= 64) nt = 2;          // one frame per call: as launch_pointwise_parts
    // 160 / 320 columns fed with fp16 pairs (the project layers behind the 576- and 960-channel depthwise results), many rows: ONE pass over
    // the operand per 160 columns in 12-wave blocks of 16 rows per wave, three operand stages in flight (tools/sweep_pwh_wide.sh at 68640 rows:
    // 960 -> 160 116 us against 139 for 128 x 80 tiles in 4-wave blocks, 576 -> 160 65 / 70, 960 -> 320 179 / 226).  What the stage loop is made
    // of (tools/sweep_pwh_abl.sh, 138 us as built): without the activation loads 68, without the weight loads 98, without either 50, without
    // MFMAs 110 — the loads do not hide behind the MFMAs, and the weight panels re-staged by every 4-wave block (330 MB at 960 x 160) weigh
    // more than the activations (264 MB): wider blocks stage them once for three times the rows, full-width tiles read the activations once
    const int epi0 = pw_pick_epi(a);
    if (a.x_fmt == 1 && a.N % 160 == 0 && a.M >= 16384 && (epi0 == EPI_PLAIN || epi0 == EPI_RES)) { rm = 1; nt = 10; }
    if (knobs().pwx_rm > 0) { rm = knobs().pwx_rm; nt = knobs().pwx_nt; }
#define PW_H(RM_, NT_) if (rm == RM_ && nt == NT_) return launch_pw_f16<RM_, NT_>(a, whi, plane, Kp, st);
    PW_H(2, 6) PW_H(2, 5) PW_H(2, 4) PW_H(2, 3) PW_H(2, 2) PW_H(2, 1)
    PW_H(1, 6) PW_H(1, 5) PW_H(1, 4) PW_H(1, 3) PW_H(1, 2) PW_H(1, 1)
    PW_H(2, 8) PW_H(2, 10) PW_H(1, 10) PW_H(4, 4) PW_H(4, 3)
#undef PW_H
    set_error("pointwise_split_f16: no tile configuration");
    return AMS_E_INVALID;
}

}  // namespace ams

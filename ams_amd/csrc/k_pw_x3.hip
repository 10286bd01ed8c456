// 1x1 convolutions, split-bf16 variant for the layers where exact f32 would be matrix-pipe bound (the output-stride-16
// section and the head): see k_pointwise.hip for the overview.
#include <type_traits>

#include "pw_common.hpp"
#include "split_bf16.hpp"

namespace ams {

// =========================================================================================================
// Split-bf16 GEMM.  f32 activations stay f32 in HBM; inside the kernel every operand is split into bf16 parts and the
// products run on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, f32 accumulate) — the f32-input MFMA runs at 1/16 of
// that rate on gfx950 and needs 8 instructions of twice the latency per 32 k:
//   NP = 3 (default, "x6"): hi + mid + lo hold all 24 significand bits; hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi,
//           6 MFMAs; the dropped terms are <= 2^-24 relative, i.e. f32 rounding level (512x1024 logits 4e-5 from the f64
//           oracle, the same as exact f32).  Inference and fine-tune step.
//   NP = 2 ("x3", opt-in for frozen inference): hi*hi + lo*hi + hi*lo, 3 MFMAs; operands cut to 16 bits, ~1e-5 per layer,
//           2e-4..5e-4 on the logits — inside the 1e-3 tolerance, not at f32 level.
// The weights are split into [N][Kp] panels (k contiguous, Kp = K rounded up to 32; parts equally spaced): once per
// ams_student_freeze for inference, per launch in the fine-tune step.
// =========================================================================================================
__device__ __forceinline__ unsigned short bf16_rne_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// parts: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid); `lo` may be null (two-part split).  Three parts hold all
// 24 significand bits of v exactly.
__global__ void split_w_kernel(const float* __restrict__ w, int64_t sk, int64_t sn, int K, int N, int Kp,
                               unsigned short* __restrict__ hi, unsigned short* __restrict__ mid, unsigned short* __restrict__ lo) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    const float v = k < K ? w[k * sk + n * sn] : 0.f;
    const unsigned short h = bf16_rne_bits(v);
    const float r1 = v - __uint_as_float((unsigned)h << 16);
    const unsigned short m = bf16_rne_bits(r1);
    hi[i] = h;
    mid[i] = m;
    if (lo) lo[i] = bf16_rne_bits(r1 - __uint_as_float((unsigned)m << 16));
}

// every live (training) weight panel of the student in ONE launch: job j owns blocks [first_block_j, first_block_{j+1})
// write_f16: the two fp16 planes of the forward jobs too (only the fine-tune step under AMS_OPT_TRAIN_FWD_F16 reads them)
__global__ void split_batch_kernel(const SplitJob* __restrict__ jobs, int njobs, int write_f16) {
    int lo = 0, hi = njobs - 1;                          // last job whose first block is <= blockIdx.x (block-uniform search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SplitJob j = jobs[lo];
    const int64_t i = ((int64_t)blockIdx.x - j.first_block) * blockDim.x + threadIdx.x;
    if (i >= (int64_t)j.N * j.Kp) return;
    const int n = (int)(i / j.Kp), k = (int)(i % j.Kp);
    const float v = k < j.K ? j.w[k * j.sk + n * j.sn] : 0.f;
    const unsigned short h = bf16_rne_bits(v);
    const float r1 = v - __uint_as_float((unsigned)h << 16);
    const unsigned short m = bf16_rne_bits(r1);
    j.p0[i] = h;
    j.p0[i + j.plane] = m;
    j.p0[i + 2 * j.plane] = bf16_rne_bits(r1 - __uint_as_float((unsigned)m << 16));
    if (j.f16 && write_f16) {
        unsigned short fh, fl;
        split1_f16(v, fh, fl);
        j.p0[i + 3 * j.plane] = fh;
        j.p0[i + 4 * j.plane] = fl;
    }
}

int launch_split_batch(const SplitJob* jobs_dev, int njobs, int64_t total_blocks, hipStream_t st, int write_f16) {
    if (njobs <= 0) return AMS_OK;
    hipLaunchKernelGGL(split_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, jobs_dev, njobs, write_f16);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_split_weights(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* lo, hipStream_t st) {
    const int64_t n = (int64_t)N * Kp;
    hipLaunchKernelGGL(split_w_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w, sk, sn, K, N, Kp, hi, lo, (unsigned short*)nullptr);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_split_weights3(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* mid, uint16_t* lo,
                          hipStream_t st) {
    const int64_t n = (int64_t)N * Kp;
    hipLaunchKernelGGL(split_w_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w, sk, sn, K, N, Kp, hi, mid, lo);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// D = stages of the operands in flight per lane (registers); the stage loop is unrolled by D so the ring is statically
// indexed, the stage count is rounded up to a multiple of D and the surplus stages only move data.  All loads are
// unconditional (clamped addresses): hipcc turns a select around a load back into an exec-masked branch, and a masked load
// costs a vmcnt(0).  D = 2 is what runs: deeper rings (4; 3 with the next stage's split interleaved into the MFMAs by
// sched_group_barrier) measured slower — per 32 k the MFMAs, the LDS fragment reads and the split VALU work add up to
// ~the measured time, and the extra registers cost a resident wave per SIMD.
// XF = PwArgs::x_mode (fine-tune step): the operand is transformed between its load and its split — BN + activation of the layer that
// wrote it (1), or the second half of that layer's BN backward from (gradient, raw output) (2) — with the per-k vectors staged in LDS once
// per block; the arithmetic is that of the elementwise pass it replaces (unfused multiply / add), so the product is bit-identical.
template <int RM, int NT, int EPI, int D, int NP, int XF = 0>
__global__ __launch_bounds__(256, (RM * NT > 12 ? 2 : (RM * NT <= 8 && NT * NP <= 15 && XF != 2) ? 4 : 3)) void pw_gemm_bf16x3_l(PwArgs a, const unsigned short* __restrict__ w0, int64_t plane,
                                                        int Kp, int n_tiles_n, unsigned nblocks, unsigned n_full) {
    // bf16 elements per LDS row: 32 k = four 16-byte pieces, piece q of row n stored at slot q ^ 2*((n >> 3) & 1).  ds_read_b128 is
    // served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): with that swap the 16 lanes of
    // a group touch 16 different 16-byte bank sets.  (The former 80-byte pitch was 2-way conflicted in every group: half of the
    // kernel's LDS cycles, profiles/r02_split_gemm_sq_counters.txt.)
    constexpr int PITCH = 32;
    constexpr int ROWS = 16 * NT;
    constexpr int NPIECE = NP * ROWS * 4;            // 16-byte pieces per stage (NP panels, 32 k = 4 pieces per row)
    constexpr int NREG = (NPIECE + 255) / 256;
    // the weight stages and the epilogue slabs share one region (the stage loop ends with a barrier): more blocks per CU
    // (+ 4 x 16 NT floats behind the slabs for the vectors of a fused column reduction, PwArgs::red_mode)
    constexpr int W_BYTES = 2 * NP * ROWS * PITCH * 2, OUT_BYTES = (EPI == EPI_GENERIC ? 4 : 4 * 16 * (16 * NT + 4) + (EPI == EPI_PLAIN ? 4 * 16 * NT : 0)) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[W_BYTES > OUT_BYTES ? W_BYTES : OUT_BYTES];
    typedef unsigned short (*WStage)[NP][ROWS * PITCH];
    WStage sW = reinterpret_cast<WStage>(smem);                                        // [buffer][part][...]
    float* sOutAll = reinterpret_cast<float*>(smem);
    __shared__ __attribute__((aligned(16))) float sSc[16 * NT], sSh[16 * NT];
    constexpr int XVS = XF != 0 ? 1024 : 4;                           // floats per operand-transform vector (K <= XVS)
    __shared__ __attribute__((aligned(16))) float sXv[XF == 1 ? 2 * XVS : XF == 2 ? 3 * XVS : 4];
    // Blocks 0 .. n_full-1 own 64*RM rows, the blocks after them 32*RM (RM/2 row groups per wave): the launcher ends a launch
    // whose last round would leave most of the chip idle with half-height tiles (pw_plan_tail).  Each section is remapped to
    // the XCDs on its own, so every XCD gets the same mix.
    const bool half = blockIdx.x >= n_full;
    const unsigned lb = half ? xcd_remap(blockIdx.x - n_full, nblocks - n_full) : xcd_remap(blockIdx.x, n_full);
    const int tile_n = lb % n_tiles_n;
    const int64_t tile_m = lb / n_tiles_n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int n0 = tile_n * ROWS;
    const int nrg = half ? RM / 2 : RM;                 // block-uniform
    const int64_t m_base = half ? (int64_t)(n_full / n_tiles_n) * (64 * RM) + tile_m * (32 * RM) + wave * (8 * RM)
                                : tile_m * (64 * RM) + wave * (16 * RM);
    const int K = a.K, n_stages = Kp / 32;
    const int n_iter = (n_stages + D - 1) / D * D;

    // vmcnt retires in order: a wait for the weight pieces of the next stage would also drain every older activation
    // load, so the weight pieces ride the same D-deep ring and are waited for at the same age.
    u32x4 wring[D][NREG];
    // this thread's weight pieces: global source (stage 0) and LDS destination, computed once — the index arithmetic is
    // loop-invariant and was a fifth of the VALU instructions of a stage
    const unsigned short* wsrc[NREG];
    int wdst[NREG];
#pragma unroll
    for (int u = 0; u < NREG; ++u) {
        const int e = tid + u * 256 < NPIECE ? tid + u * 256 : NPIECE - 1;         // surplus lanes repeat the last piece
        const int which = e / (ROWS * 4), r = e - which * (ROWS * 4), n = r >> 2, part = r & 3;
        int nn = n0 + n;
        if (nn > a.N - 1) nn = a.N - 1;
        // columns >= N repeat column N-1: they are never stored.  No select on the loaded value: hipcc would turn it back
        // into an exec-masked branch around the load, and a masked load costs a vmcnt(0).  Part p of the panels starts at
        // w0 + p * plane (one base pointer: a select between pointers becomes a stack table).
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

    const float* arow[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        int64_t m = m_base + (r < nrg ? r : nrg - 1) * 16 + l15;        // an unused row group repeats the last used one (L1 hits)
        if (m > a.M - 1) m = a.M - 1;
        arow[r] = a.x + m * (int64_t)a.ldx;
    }
    float4 abuf[D][RM][2];
    float4 zbuf[XF == 2 ? D : 1][XF == 2 ? RM : 1][2];       // x_mode 2: the raw output z beside the gradient, same ring
    const int64_t x2_off = XF == 2 ? (int64_t)(a.x2 - a.x) : 0;
    auto load_a = [&](int s, float4 (&dst)[RM][2], float4 (&dz)[XF == 2 ? RM : 1][2], auto Rc) {
        constexpr int R = decltype(Rc)::value;
        // k >= K repeats the last 8 k of the row: the weight panels are zero there (split_w_kernel pads to Kp)
        int koff = s * 32 + 8 * q;
        if (koff > K - 8) koff = K - 8;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dst[r][0] = ld4(arow[r] + koff);
            dst[r][1] = ld4(arow[r] + koff + 4);
            if constexpr (XF == 2) {
                dz[r][0] = ld4(arow[r] + x2_off + koff);
                dz[r][1] = ld4(arow[r] + x2_off + koff + 4);
            }
        }
    };
    typedef std::integral_constant<int, RM> RFull;
    typedef std::integral_constant<int, (RM >= 2 ? RM / 2 : RM)> RHalf;
    load_stage(0, wring[0]);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
        if (d > 0) load_stage(d, wring[d]);
        load_a(d, abuf[d], zbuf[XF == 2 ? d : 0], RFull{});
    }
    f32x4 acc[RM][NT];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    pw_stage_affine<NT>(a, sSc, sSh, n0, tid, 256);
    if constexpr (XF != 0) {
        for (int e = tid; e < XVS; e += 256) {
            const bool ok = e < K;
            sXv[e] = ok ? a.x_v0[e] : 0.f;
            sXv[XVS + e] = ok ? a.x_v1[e] : 0.f;
            if constexpr (XF == 2) sXv[2 * XVS + e] = ok ? a.x_v2[e] : 0.f;
        }
    }
    store_stage(0, wring[0]);
    __syncthreads();
    auto main_loop = [&](auto Rc) {
    constexpr int R = decltype(Rc)::value;           // row groups of this wave that exist (RM, or RM / 2 in a half-height block)
    for (int s0 = 0; s0 < n_iter; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int s = s0 + d;
            load_stage(s + D - 1, wring[(d + D - 1) % D]);
            load_a(s + D - 1, abuf[(d + D - 1) % D], zbuf[XF == 2 ? (d + D - 1) % D : 0], Rc);
            if (s < n_stages) {                       // block-uniform: surplus stages of the rounded-up loop only move data
                bf16x8 x0[RM], x1[RM], x2[RM];
                if constexpr (XF != 0) {
                    int koff = s * 32 + 8 * q;
                    if (koff > K - 8) koff = K - 8;
                    const float4 u0 = ld4(sXv + koff), u1 = ld4(sXv + koff + 4), v0 = ld4(sXv + XVS + koff), v1 = ld4(sXv + XVS + koff + 4);
                    if constexpr (XF == 1) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float4 y0 = muladd4_pk(abuf[d][r][0], u0, v0), y1 = muladd4_pk(abuf[d][r][1], u1, v1);
                            y0 = make_float4(apply_act(y0.x, a.x_act), apply_act(y0.y, a.x_act), apply_act(y0.z, a.x_act), apply_act(y0.w, a.x_act));
                            y1 = make_float4(apply_act(y1.x, a.x_act), apply_act(y1.y, a.x_act), apply_act(y1.z, a.x_act), apply_act(y1.w, a.x_act));
                            split8(y0, y1, x0[r], x1[r], x2[r]);
                        }
                    } else {
                        const float4 c0 = ld4(sXv + 2 * XVS + koff), c1 = ld4(sXv + 2 * XVS + koff + 4);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            // (A g + B) + C z, as bn_bwd_apply_kernel evaluates it
                            const float4 y0 = add4_pk(add4_pk(mul4_pk(u0, abuf[d][r][0]), v0), mul4_pk(c0, zbuf[XF == 2 ? d : 0][r][0]));
                            const float4 y1 = add4_pk(add4_pk(mul4_pk(u1, abuf[d][r][1]), v1), mul4_pk(c1, zbuf[XF == 2 ? d : 0][r][1]));
                            split8(y0, y1, x0[r], x1[r], x2[r]);
                        }
                    }
                } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (NP == 3) split8(abuf[d][r][0], abuf[d][r][1], x0[r], x1[r], x2[r]);
                    else if (NP == 2) split8(abuf[d][r][0], abuf[d][r][1], x0[r], x1[r]);
                    else split8(abuf[d][r][0], abuf[d][r][1], x0[r]);
                }
                }
                const unsigned short* bw = &sW[d & 1][0][l15 * PITCH + 8 * (q ^ ((l15 >> 3) << 1))];
                // Per accumulator the six products arrive in a fixed order (smallest terms first); consecutive MFMAs rotate over the
                // TG * RM accumulators of a group.  (The rotation is not needed against a dependency stall — a bf16 MFMA on its
                // predecessor's accumulator issues at the pipe rate, tools/probes/mfma_chain_probe.hip — but issuing an accumulator's six
                // products back to back measured no faster here and slower in the streaming kernels, where the next accumulator's first
                // product then waits for its LDS fragments with nothing else to issue.)  TG * NP weight fragments live.
                constexpr int TG = (NT >= 2 && RM <= 2) ? 2 : 1;        // an odd NT ends with a group of one
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += TG) {
                    bf16x8 q0[TG], q1[TG], q2[TG];
#pragma unroll
                    for (int g = 0; g < TG; ++g) {
                        if (t0 + g >= NT) continue;
                        q0[g] = *reinterpret_cast<const bf16x8*>(bw + (t0 + g) * 16 * PITCH);
                        if (NP >= 2) q1[g] = *reinterpret_cast<const bf16x8*>(bw + (ROWS * PITCH) + (t0 + g) * 16 * PITCH);
                        if (NP == 3) q2[g] = *reinterpret_cast<const bf16x8*>(bw + 2 * (ROWS * PITCH) + (t0 + g) * 16 * PITCH);
                    }
#define AMS_X3_TERM(QA, XB)                                                                                          \
    _Pragma("unroll") for (int g = 0; g < TG; ++g)                                                                   \
        _Pragma("unroll") for (int r = 0; r < R; ++r)                                                                \
            if (t0 + g < NT)                                                                                         \
                acc[r][t0 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QA[g], XB[r], acc[r][t0 + g], 0, 0, 0);
                    if (NP == 3) {           // smallest terms first
                        AMS_X3_TERM(q2, x0)
                        AMS_X3_TERM(q0, x2)
                        AMS_X3_TERM(q1, x1)
                    }
                    if (NP >= 2) {
                        AMS_X3_TERM(q1, x0)
                        AMS_X3_TERM(q0, x1)
                    }
                    AMS_X3_TERM(q0, x0)
#undef AMS_X3_TERM
                }
            }
            store_stage((d + 1) & 1, wring[(d + 1) % D]);
            __syncthreads();
        }
    }
    };
    if (RM >= 2 && half) main_loop(RHalf{});
    else main_loop(RFull{});
    const bool red = EPI == EPI_PLAIN && a.red_mode != 0;           // block-uniform (the launcher clears red_mode for the other epilogues)
    float4 rs1[EPI == EPI_PLAIN ? NT : 1], rs2[EPI == EPI_PLAIN ? NT : 1];
    if constexpr (EPI == EPI_PLAIN) {
        if (red) {
            // the stage loop ended with a barrier: the weight stages are dead, their LDS holds the reduction's vectors and, later, the waves' sums
            float* sRedVec = reinterpret_cast<float*>(smem) + 4 * 16 * (16 * NT + 4);
            pw_red_stage<NT>(a, sRedVec, n0, tid, 256);
#pragma unroll
            for (int t = 0; t < NT; ++t) { rs1[t] = make_float4(0.f, 0.f, 0.f, 0.f); rs2[t] = rs1[t]; }
            __syncthreads();
            pw_red_rowgroups<RM, NT>(a, acc, m_base, n0, l15, q, sRedVec, rs1, rs2, nrg);
        }
    }
    if (EPI == EPI_GENERIC) pw_epilogue<RM, NT>(a, acc, m_base, n0, l15, q, sSc, sSh, nrg);
    else pw_epilogue_t<RM, NT, EPI, true>(a, acc, m_base, n0, lane, sSc, sSh, sOutAll + wave * (16 * (16 * NT + 4)), nrg);
    if constexpr (EPI == EPI_PLAIN) {
        if (red) {
            __syncthreads();                                        // the epilogue slabs are consumed
            const int64_t row = half ? (int64_t)(n_full / n_tiles_n) + tile_m : tile_m;
            pw_red_finish<NT>(a, rs1, rs2, lane, wave, 4, reinterpret_cast<float*>(smem), row, n0, tid, 256);
        }
    }
}

struct SplitPanels { const uint16_t* base; int64_t plane; int np; };     // part p at base + p * plane

// One frame per call (M <= 2400 rows): the launch is one block's chain of K / 32 stages (~0.33 us each, 10 us for K = 960 whatever the
// width).  Two other forms of that chain were built and measured in round 4, both bit-identical and both SLOWER: three stages in flight
// instead of one (0.505 vs 0.481 ms per frame), and a kernel without LDS or barriers in which every wave owns a 16 x 32 tile and loads its
// weight fragments straight from the panels (0.722 ms: 64-byte row pieces from panels that are not L2-resident between frames cost more
// latency than two stages of prefetch cover, and deeper rings do not fit the registers).

template <int RM, int NT, int EPI, int D, int NP, int XF = 0>
static int launch_pw_x3_d(const PwArgs& a, const SplitPanels& w, int Kp, hipStream_t st) {
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    int per_cu = 1, cus = 256;                     // resident blocks of this instantiation on the whole chip (per device)
    RUN_RC(func_blocks_per_cu((const void*)pw_gemm_bf16x3_l<RM, NT, EPI, D, NP, XF>, 256, 0, &per_cu));
    RUN_RC(device_cus(&cus));
    const int slots = per_cu * cus;
    int64_t half_strips = 0;
    const int64_t full_strips = pw_plan_tail(a.M, RM, n_tiles_n, slots, &half_strips);
    const int64_t n_full = full_strips * n_tiles_n;
    const int64_t nblocks = n_full + half_strips * n_tiles_n;
    PwArgs b = a;
    if (b.red_mode) {
        if (EPI == EPI_PLAIN && pw_red_ok(b, nblocks / n_tiles_n)) { if (b.red_rows_out) *b.red_rows_out = (int)(nblocks / n_tiles_n); }
        else { b.red_mode = 0; if (b.red_rows_out) *b.red_rows_out = 0; }
    }
    // the kernel's own symbol (rocprofv3 reports the same text); NP = 3 is the six-product "x6" training variant
    static const std::string nm = "pw_gemm_bf16x3_l<" + std::to_string(RM) + ", " + std::to_string(NT) + ", " + std::to_string(EPI) +
                                  ", " + std::to_string(D) + ", " + std::to_string(NP) + (XF ? ", " + std::to_string(XF) : std::string()) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_gemm_bf16x3_l<RM, NT, EPI, D, NP, XF>), dim3((unsigned)nblocks), dim3(256), 0, st, b, w.base, w.plane, Kp,
                       n_tiles_n, (unsigned)nblocks, (unsigned)n_full);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// the operand transform exists for the fine-tune step's form only: three parts, plain epilogue, K within the LDS vectors
static bool pw_x3_xform_ok(const PwArgs& a, int np, int epi) {
    return np == 3 && epi == EPI_PLAIN && a.x_v0 && a.x_v1 &&
           ((a.x_mode == 1 && a.K <= 1024) || (a.x_mode == 2 && a.K <= 1024 && a.x_v2 && a.x2 && a.x_act == AMS_ACT_NONE));
}

template <int RM, int NT, int EPI>
static int launch_pw_x3_e(const PwArgs& a, const SplitPanels& w, int Kp, hipStream_t st) {
    if (a.x_mode != 0) {
        if constexpr (EPI == EPI_PLAIN) {
            if (pw_x3_xform_ok(a, w.np, EPI))
                return a.x_mode == 1 ? launch_pw_x3_d<RM, NT, EPI_PLAIN, 2, 3, 1>(a, w, Kp, st) : launch_pw_x3_d<RM, NT, EPI_PLAIN, 2, 3, 2>(a, w, Kp, st);
        }
        PwArgs b;
        RUN_RC(pointwise_materialize_x(a, &b, st));
        return launch_pw_x3_e<RM, NT, EPI>(b, w, Kp, st);
    }
    // D = 4 is no faster (measured): at full occupancy the stage loop is bound by the LDS hand-over of the weight pieces, not by latency; and
    // at one frame per call (at most one block per CU, the launch = one block's chain of K / 32 stages) three stages in flight are SLOWER
    // (0.505 vs 0.481 ms per frame, round 4): the chain is the barrier + LDS hand-over per stage, not the L2 round trip
    if (w.np == 3) return launch_pw_x3_d<RM, NT, EPI, 2, 3>(a, w, Kp, st);
    if (w.np == 1) return launch_pw_x3_d<RM, NT, EPI, 2, 1>(a, w, Kp, st);
    return launch_pw_x3_d<RM, NT, EPI, 2, 2>(a, w, Kp, st);
}

template <int RM, int NT>
static int launch_pw_x3(const PwArgs& a, const SplitPanels& w, int Kp, hipStream_t st) {
    if (a.red_mode == 2 && a.res && !a.img_bias && (a.ldr & 3) == 0) {
        // dgrad GEMM whose result also takes the residual branch's gradient: the fused reduction adds it (acc + res, what EPI_RES
        // computes) ahead of the mask and the sums, and the plain epilogue stores the total
        PwArgs b = a;
        b.red_res = a.res; b.red_ldr = a.ldr; b.res = nullptr;
        if (pw_red_ok(b, cdiv64(b.M, 32) + 2)) return launch_pw_x3_e<RM, NT, EPI_PLAIN>(b, w, Kp, st);      // (upper bound of the strips: the exact count is checked again at launch)
    }
    switch (pw_pick_epi(a)) {
        case EPI_PLAIN: return launch_pw_x3_e<RM, NT, EPI_PLAIN>(a, w, Kp, st);
        case EPI_RES: return launch_pw_x3_e<RM, NT, EPI_RES>(a, w, Kp, st);
        case EPI_BIAS: return launch_pw_x3_e<RM, NT, EPI_BIAS>(a, w, Kp, st);
        default: return launch_pw_x3_e<RM, NT, EPI_GENERIC>(a, w, Kp, st);
    }
}

static int launch_pointwise_parts(const PwArgs& a, const SplitPanels& w, int Kp, hipStream_t st) {
    if (a.red_rows_out) *a.red_rows_out = 0;               // set below when the launch fuses the column reduction (PwArgs::red_mode)
    AMS_REQUIRE(a.M > 0 && a.K > 0 && a.N > 0 && Kp % 32 == 0 && Kp >= a.K, "pointwise_split: bad problem");
    AMS_REQUIRE(a.K % 8 == 0 && a.ldx % 4 == 0, "pointwise_split: K (%d) must be a multiple of 8", a.K);
    int rm, nt;
    pw_pick_tile(a.M, a.N, &rm, &nt);
    // second sweep for this kernel (tools/sweep_pwx.sh, three-part split, 68640 rows): a 96-wide tile covers the 96-column layers
    // in one pass over the operand (+10 %), 160-wide tiles the 320-column layer in two (+8 %), 960 columns run best as ten
    // 96-wide tiles (+8 %).  Only with two row groups per wave (many rows): the one-row-group variants keep the narrow tiles.
    if (rm == 2) {
        if (a.N == 96 || a.N == 960) nt = 6;
        else if (a.N == 320) nt = 10;
    }
    // one frame per call (33 x 65 = 2145 rows at the output stride): 34 row strips cannot fill the chip with 64- or 80-wide column tiles;
    // 32-wide tiles give 2-5x the blocks for the same k loop (same products in the same order: same bits).  0.495 -> 0.480 ms per frame;
    // from two frames on the wide tiles win again (measured at 2, 3, 4 frames)
    if (a.M <= 2400 && rm == 1 && a.N % 32 == 0 && a.N >= 64) nt = 2;
    if (knobs().pwx_rm > 0) { rm = knobs().pwx_rm; nt = knobs().pwx_nt; }            // tuning knob AMS_PWX_FORCE
#define PW_X(RM_, NT_) if (rm == RM_ && nt == NT_) return launch_pw_x3<RM_, NT_>(a, w, Kp, st);
    PW_X(4, 4) PW_X(4, 3) PW_X(4, 5)
    PW_X(2, 10) PW_X(2, 8)
    PW_X(2, 6) PW_X(2, 5) PW_X(2, 4) PW_X(2, 3) PW_X(2, 2) PW_X(2, 1)
    PW_X(1, 6) PW_X(1, 5) PW_X(1, 4) PW_X(1, 3) PW_X(1, 2) PW_X(1, 1)
#undef PW_X
    set_error("pointwise_split: no tile configuration");
    return AMS_E_INVALID;
}

// the three-part launch of this problem applies PwArgs::x_mode on its operand loads (else it materialises x' into x_tmp first)
bool pointwise_split3_transforms_on_load(const PwArgs& a) {
    if (a.x_mode == 0) return true;
    return pw_pick_epi(a) == EPI_PLAIN && pw_x3_xform_ok(a, 3, EPI_PLAIN);
}

bool pointwise_split_writes_parts(const PwArgs& a) { return pw_pick_epi(a) != EPI_GENERIC && a.N % 4 == 0; }

// y = epilogue(x @ w) with w given as pre-split bf16 hi/lo panels [N][Kp]; requires K % 8 == 0
int launch_pointwise_split(const PwArgs& a, const uint16_t* whi, const uint16_t* wlo, int Kp, hipStream_t st) {
    const SplitPanels w = {whi, (int64_t)(wlo - whi), 2};
    return launch_pointwise_parts(a, w, Kp, st);
}

// one part: plain bf16 products (AMS_MATMUL_BF16, the opt-in bf16 inference variant): 1 MFMA per 32 k
int launch_pointwise_split1(const PwArgs& a, const uint16_t* whi, int Kp, hipStream_t st) {
    const SplitPanels w = {whi, 0, 1};
    return launch_pointwise_parts(a, w, Kp, st);
}

// the same with three-part panels (hi, mid, lo): f32-level accuracy at 6 bf16 MFMAs per 32 k
int launch_pointwise_split3(const PwArgs& a, const uint16_t* whi, const uint16_t* wmid, const uint16_t* wlo, int Kp, hipStream_t st) {
    AMS_REQUIRE(wlo - wmid == wmid - whi, "pointwise_split3: the three panels must be equally spaced");
    const SplitPanels w = {whi, (int64_t)(wmid - whi), 3};
    return launch_pointwise_parts(a, w, Kp, st);
}

}  // namespace ams

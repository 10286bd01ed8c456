// Fused depthwise 3x3 (BN, ReLU6) -> project 1x1 (BN, + residual) of an inverted-residual block, frozen inference, for the
// output-stride-16 blocks (few pixels, 384..960 expanded channels).
//
// Layer by layer the depthwise output d (6x the block width) is written by one kernel and read back by the next:
// 2 x 26..66 MB per block and step, plus a launch.  Here the project GEMM produces its own operand: it is the split-bf16
// GEMM of k_pw_x3.hip whose activation fragment is not loaded but COMPUTED — in the MFMA operand layout a lane owns 8
// consecutive expanded channels of one pixel, so it reads the 9 taps of exactly those from the LDS input tile, applies
// the depthwise weights, BN and ReLU6 in registers, splits the result into bf16 hi/lo and feeds the matrix pipe.  d never exists in memory.  Arithmetic order of the depthwise sum and of the GEMM is the same as in the
// separate kernels.
//
// Block = 4 waves = 4 image rows x 16 columns (wave w: row 4*ty + w); all 16*NT output channels of one column tile.
// Per 32-channel chunk the block stages in LDS: the input tile incl. halo ((4+2R) x (16+2R) pixels, zeros outside the
// image — loading the 9 taps per lane instead costs 72 registers of prefetch and 5x the L2 traffic), the project-weight
// pieces and the depthwise weights / BN of the chunk (single buffers, two barriers per chunk: up to three blocks per CU).
#include "pw_common.hpp"

namespace ams {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct DwpArgs {
    const float* e;          // expanded activations [B, H, W, C]
    int B, H, W, C;          // C = expanded channels (multiple of 32)
    const float* w_dw;       // [9][C]
    const float* sc_d; const float* sh_d;       // folded BN of the depthwise layer
    int act_d, rate;
    PwArgs p;                // project layer: K = C, N, scale, shift, act, res, ldr, y, ldy (x, M unused / set by the launcher)
    const unsigned short* whi;                  // project weights split into bf16 panels [N][Kp]: hi at whi, lo at whi + plane
    int64_t plane;
    int tiles_x, tiles_y, n_tiles_n;
};

__device__ __forceinline__ void dwp_split8(const float (&f)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        hi[j] = h;
        lo[j] = (__bf16)(f[j] - (float)h);
    }
}

template <int NT, int R>
__global__ __launch_bounds__(256, 2) void dw_project_kernel(DwpArgs a, unsigned nblocks) {
    constexpr int PITCH = 40;                         // bf16 per LDS row of the weight stage (80 B: conflict-free b128 reads)
    constexpr int ROWS = 16 * NT;
    constexpr int NPIECE = 2 * ROWS * 4;              // 16-byte pieces of one stage (hi + lo, 32 k = 4 pieces per row)
    constexpr int NREG = (NPIECE + 255) / 256;
    constexpr int WSTAGE = 2 * ROWS * PITCH;          // unsigned shorts
    constexpr int IH = 4 + 2 * R, IW = 16 + 2 * R;    // input tile incl. halo
    constexpr int TP = 36;                            // floats per tile pixel: 32 channels + 4 (pixel stride 144 B: the 16 lanes
                                                      // of a b128 read pass land in 16 different bank quads)
    constexpr int NTV = IH * IW * 8;                  // float4 of one tile chunk
    constexpr int NTREG = (NTV + 255) / 256;
    constexpr int TSTAGE = IH * IW * TP;              // floats
    constexpr int PSTAGE = 11 * 32;                   // floats: 9 taps + scale + shift for 32 channels
    constexpr int OUT_FLOATS = 4 * 16 * (16 * NT + 4);
    constexpr int LOOP_BYTES = WSTAGE * 2 + (TSTAGE + PSTAGE) * 4;
    constexpr int SMEM_BYTES = LOOP_BYTES > OUT_FLOATS * 4 ? LOOP_BYTES : OUT_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];        // stage buffers, then the epilogue slabs
    __shared__ __attribute__((aligned(16))) float sSc[16 * NT], sSh[16 * NT];
    unsigned short* sW = reinterpret_cast<unsigned short*>(smem);                  // [2][ROWS*PITCH]   (hi, lo)
    float* sT = reinterpret_cast<float*>(smem + WSTAGE * 2);                       // [IH][IW][TP]
    float* sP = sT + TSTAGE;                                                       // [11][32]

    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int tile_n = lb % a.n_tiles_n;
    unsigned t1 = lb / a.n_tiles_n;
    const int tx = t1 % a.tiles_x; t1 /= a.tiles_x;
    const int ty = t1 % a.tiles_y;
    const int b = t1 / a.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = tile_n * ROWS;
    const int y0 = ty * 4, x0 = tx * 16;
    const int n_stages = a.C / 32;

    // ---- per-thread pieces of a 32-channel chunk: input tile (with halo; zero outside the image = SAME padding), project
    // weight pieces, depthwise parameters.  Addresses and validity do not depend on the chunk.
    const float* eb = a.e + (int64_t)b * a.H * a.W * a.C;
    int toff[NTREG];                                   // element offset of this thread's float4 in chunk 0 (clamped)
    unsigned tok = 0;
#pragma unroll
    for (int u = 0; u < NTREG; ++u) {
        const int e = tid + 256 * u < NTV ? tid + 256 * u : NTV - 1;
        const int px = e >> 3, c4 = (e & 7) * 4;
        const int iy = px / IW, ix = px - iy * IW;
        const int yy = y0 - R + iy, xx = x0 - R + ix;
        const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        const int yc = yy < 0 ? 0 : (yy > a.H - 1 ? a.H - 1 : yy), xc = xx < 0 ? 0 : (xx > a.W - 1 ? a.W - 1 : xx);
        toff[u] = (yc * a.W + xc) * a.C + c4;
        tok |= ok ? (1u << u) : 0u;
    }
    // All pieces ride a ring of three register sets and are requested TWO chunks ahead: with one chunk of lead the loop is
    // a chain of 12..30 dependent memory round trips, and at ~1.4 blocks per CU nothing else hides them.
    float4 tring[3][NTREG];
    u32x4 wring[3][NREG];
    float4 pring[3];
    auto load_tile = [&](int s, float4 (&treg)[NTREG]) {
        if (s > n_stages - 1) s = n_stages - 1;
#pragma unroll
        for (int u = 0; u < NTREG; ++u) treg[u] = ld4(eb + toff[u] + s * 32);
    };
    auto load_w = [&](int s, u32x4 (&wreg)[NREG], float4& preg) {
        if (s > n_stages - 1) s = n_stages - 1;
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int e = tid + u * 256 < NPIECE ? tid + u * 256 : NPIECE - 1;
            const int which = e / (ROWS * 4), r = e - which * (ROWS * 4), n = r >> 2, part = r & 3;
            int nn = n0 + n;
            if (nn > a.p.N - 1) nn = a.p.N - 1;
            wreg[u] = *reinterpret_cast<const u32x4*>(a.whi + which * a.plane + (int64_t)nn * a.p.K + s * 32 + part * 8);
        }
        if (tid < 88) {                                // 11 rows x 8 float4
            const int row = tid >> 3, c4 = (tid & 7) * 4;
            const float* src = row < 9 ? a.w_dw + (int64_t)row * a.C : (row == 9 ? a.sc_d : a.sh_d);
            preg = ld4(src + s * 32 + c4);
        }
    };
    auto store_chunk = [&](const float4 (&treg)[NTREG], const u32x4 (&wreg)[NREG], const float4& preg) {
#pragma unroll
        for (int u = 0; u < NTREG; ++u) {
            const int e = tid + 256 * u;
            if (e < NTV) {
                const bool ok = (tok >> u) & 1u;
                const float4 v = treg[u];
                st4(sT + (e >> 3) * TP + (e & 7) * 4, make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f));
            }
        }
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int e = tid + u * 256 < NPIECE ? tid + u * 256 : NPIECE - 1;
            const int which = e / (ROWS * 4), r = e - which * (ROWS * 4), n = r >> 2, part = r & 3;
            *reinterpret_cast<u32x4*>(&sW[which * (ROWS * PITCH) + n * PITCH + part * 8]) = wreg[u];
        }
        if (tid < 88) st4(sP + (tid >> 3) * 32 + (tid & 7) * 4, preg);
    };

    f32x4 acc[1][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    pring[0] = pring[1] = pring[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_w(0, wring[0], pring[0]);
    load_tile(0, tring[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_w(1, wring[1], pring[1]);
    load_tile(1, tring[1]);
    pw_stage_affine<NT>(a.p, sSc, sSh, n0, tid, 256);
    store_chunk(tring[0], wring[0], pring[0]);
    __syncthreads();
    // this lane's pixel inside the tile: row `wave`, column l15; its 8 channels start at 8 q
    const float* tp = sT + ((wave + R) * IW + (l15 + R)) * TP + 8 * q;
    const float* pp = sP + 8 * q;
    const int n_iter = (n_stages + 2) / 3 * 3;
    for (int s0 = 0; s0 < n_iter; s0 += 3) {
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
            const int s = s0 + dd;
            load_w(s + 2, wring[(dd + 2) % 3], pring[(dd + 2) % 3]);
            load_tile(s + 2, tring[(dd + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (s < n_stages) {                        // block-uniform (the rounded-up tail only moves data)
                float d[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) d[c] = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float* tq = tp + ((i - 1) * R * IW + (j - 1) * R) * TP;
                        const float4 v0 = ld4(tq), v1 = ld4(tq + 4);
                        const float4 w0 = ld4(pp + (i * 3 + j) * 32), w1 = ld4(pp + (i * 3 + j) * 32 + 4);
                        d[0] = fmaf(v0.x, w0.x, d[0]); d[1] = fmaf(v0.y, w0.y, d[1]); d[2] = fmaf(v0.z, w0.z, d[2]); d[3] = fmaf(v0.w, w0.w, d[3]);
                        d[4] = fmaf(v1.x, w1.x, d[4]); d[5] = fmaf(v1.y, w1.y, d[5]); d[6] = fmaf(v1.z, w1.z, d[6]); d[7] = fmaf(v1.w, w1.w, d[7]);
                    }
                {
                    const float4 c0 = ld4(pp + 9 * 32), c1 = ld4(pp + 9 * 32 + 4), h0 = ld4(pp + 10 * 32), h1 = ld4(pp + 10 * 32 + 4);
                    d[0] = apply_act(d[0] * c0.x + h0.x, a.act_d); d[1] = apply_act(d[1] * c0.y + h0.y, a.act_d);
                    d[2] = apply_act(d[2] * c0.z + h0.z, a.act_d); d[3] = apply_act(d[3] * c0.w + h0.w, a.act_d);
                    d[4] = apply_act(d[4] * c1.x + h1.x, a.act_d); d[5] = apply_act(d[5] * c1.y + h1.y, a.act_d);
                    d[6] = apply_act(d[6] * c1.z + h1.z, a.act_d); d[7] = apply_act(d[7] * c1.w + h1.w, a.act_d);
                }
                bf16x8 xh, xl;
                dwp_split8(d, xh, xl);
                const unsigned short* bh = sW + l15 * PITCH + 8 * q;
                const unsigned short* bl = bh + ROWS * PITCH;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(bh + t * 16 * PITCH);
                    const bf16x8 wl = *reinterpret_cast<const bf16x8*>(bl + t * 16 * PITCH);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[0][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[0][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[0][t], 0, 0, 0);
                }
            }
            __syncthreads();                           // every wave is done reading this chunk's tile and weights
            store_chunk(tring[(dd + 1) % 3], wring[(dd + 1) % 3], pring[(dd + 1) % 3]);
            __syncthreads();
        }
    }
    // ---- epilogue: the wave's 16 pixels are consecutive rows of the [B*H*W, N] output; columns beyond the image row are
    // cut off through the row limit.  The stage buffers are dead now and become the per-wave slabs.
    PwArgs pw = a.p;
    const int y = y0 + wave;
    const int64_t m0 = ((int64_t)b * a.H + y) * a.W + x0;
    int valid = a.W - x0;
    valid = valid > 16 ? 16 : valid;
    if (y >= a.H) valid = 0;
    pw.M = m0 + valid;
    float* sOut = reinterpret_cast<float*>(smem) + wave * (16 * (16 * NT + 4));
    if (pw.res) pw_epilogue_t<1, NT, EPI_RES>(pw, acc, m0, n0, lane, sSc, sSh, sOut);
    else pw_epilogue_t<1, NT, EPI_PLAIN>(pw, acc, m0, n0, lane, sSc, sSh, sOut);
}

static int dwp_pick_nt(int N) {
    const int n16 = N / 16;
    const int nt = n16 <= 10 ? n16 : (n16 % 10 == 0 ? 10 : 8);
    return (nt >= 1 && nt <= 6) || nt == 8 || nt == 10 ? ((n16 % nt == 0) ? nt : 0) : 0;
}

bool dw_project_supported(int C, int N, int stride, int rate) {
    return stride == 1 && (rate == 1 || rate == 2) && C % 32 == 0 && N % 16 == 0 && dwp_pick_nt(N) > 0;
}

template <int NT, int R>
static int launch_dwp_t(DwpArgs a, hipStream_t st) {
    a.tiles_x = cdiv(a.W, 16);
    a.tiles_y = cdiv(a.H, 4);
    a.n_tiles_n = cdiv(a.p.N, 16 * NT);
    const int64_t nblocks = (int64_t)a.tiles_x * a.tiles_y * a.B * a.n_tiles_n;
    AMS_REQUIRE(nblocks > 0 && nblocks < 0x7fffffffLL, "dw_project: bad grid");
    static const std::string nm = "dw_project_kernel<" + std::to_string(NT) + ", " + std::to_string(R) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((dw_project_kernel<NT, R>), dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// e: [B,H,W,C] -> y: [B,H,W,N]; p carries the project layer (N, scale, shift, act, res, ldr, y, ldy); whi/wlo: its
// weights as bf16 panels [N][C]
int launch_dw_project(const float* e, int B, int H, int W, int C, const float* w_dw, int rate, const float* sc_d, const float* sh_d,
                      int act_d, const PwArgs& p, const uint16_t* whi, const uint16_t* wlo, int Kp, hipStream_t st) {
    AMS_REQUIRE(dw_project_supported(C, p.N, 1, rate), "dw_project: unsupported shape C=%d N=%d rate=%d", C, p.N, rate);
    AMS_REQUIRE(Kp == C && p.K == C, "dw_project: the panels must be [N][C] (Kp=%d, K=%d, C=%d)", Kp, p.K, C);
    AMS_REQUIRE((p.ldy & 3) == 0 && (!p.res || (p.ldr & 3) == 0) && !p.img_bias, "dw_project: layout");
    DwpArgs a;
    memset(&a, 0, sizeof(a));
    a.e = e; a.B = B; a.H = H; a.W = W; a.C = C; a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.rate = rate;
    a.p = p; a.whi = whi; a.plane = (int64_t)(wlo - whi);
    const int nt = dwp_pick_nt(p.N);
#define DWP(NT_) if (nt == NT_) return rate == 1 ? launch_dwp_t<NT_, 1>(a, st) : launch_dwp_t<NT_, 2>(a, st);
    DWP(1) DWP(2) DWP(3) DWP(4) DWP(5) DWP(6) DWP(8) DWP(10)
#undef DWP
    set_error("dw_project: no instantiation for N=%d", p.N);
    return AMS_E_INVALID;
}

}  // namespace ams

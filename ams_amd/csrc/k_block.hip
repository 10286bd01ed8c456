// One whole inverted-residual block of the early (high-resolution) section in one kernel, frozen inference:
//   expand 1x1 + BN + ReLU6  ->  depthwise 3x3 (stride 1 | 2) + BN + ReLU6  ->  project 1x1 + BN (+ block input)
// Only the block input and the block output touch HBM: neither the 6x-expanded tensor nor the depthwise result `d` is ever
// written (SURVEY §8 d4 "block-fused convention").  In the layer-by-layer plan `d` alone is 1.6 GB of traffic per 32 frames over
// blocks 1-6, and the five early project GEMMs that read it back cost 0.47 ms of a 4.5 ms step.
//
// A block owns a TH x TW tile of OUTPUT pixels and walks all channel chunks (16*NT expanded channels each) of it:
//   phase 1  exact-f32 MFMA GEMM over the tile's input pixels incl. the 3x3 halo, BN + ReLU6, into LDS (k_expand_dw.hip's phase 1:
//            operands gathered once per tile, zeros outside the image = the depthwise conv's SAME padding);
//   phase 2  depthwise 3x3 from LDS, BN + ReLU6 — per WAVE and 16-pixel output row group, lane (pixel l15, k-group q) forming the
//            four channels 16kc + 4q .. +3 of its pixel: exactly the lane's operand of
//   phase 3  the project MFMAs (exact f32, roles swapped: a lane owns 4 consecutive output channels of one pixel), whose
//            accumulators live across the chunks: `d` never leaves the registers.
// The weights of a chunk are MFMA operand A and live in registers as well (KC*4*NT + NT*4*NTO floats per lane, requested one
// chunk ahead straight from L2); LDS holds the expanded tile of one chunk and the per-channel vectors of all chunks, staged once:
// two barriers per chunk, 40-60 KB per block.
// After the last chunk: BN of the project layer, residual (= the block input at the output pixel), float4 stores.
// Products, k order and tap order are those of the kernels it replaces (pw_gemm_f32_s walks k in 16-wide chunks, 4 MFMA steps
// each, k = 16c + 4q + j; so does phase 3 across the channel chunks): the result is bit-identical to the layer-by-layer plan.
#include "pw_common.hpp"

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
};

template <int S, int NT, int TH, int TW, int KC, int NTO>
__global__ __launch_bounds__(256) void block_kernel(BlkArgs a, unsigned nblocks) {
    constexpr int CC = 16 * NT;                       // expanded channels per chunk
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int NPIX = IH * IW;
    constexpr int NRG = (NPIX + 15) / 16;             // 16-pixel row groups of the input tile
    constexpr int MRG = (NRG + 3) / 4;                // input row groups per wave
    constexpr int AP = CC + 4;
    constexpr int ORG = TH * TW / 16;                 // output row groups of the tile
    static_assert(TH * TW % 64 == 0, "every wave takes the same number of output row groups");
    constexpr int MRO = ORG / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // per-channel vectors of ALL chunks, staged once: sc_e | sh_e | sc_d | sh_d | depthwise taps [9][Cexp]
    float* sVec = smem;                               // [13][Cexp]
    float* sAct = smem + 13 * a.Cexp;                 // [NRG*16][AP]    expanded tile incl. halo, one chunk at a time

    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    unsigned t1 = lb;
    const int tx = t1 % a.tiles_x; t1 /= a.tiles_x;
    const int ty = t1 % a.tiles_y;
    const int b = t1 / a.tiles_y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;

    // ---- operands of this wave's input row groups, requested up front (clamped, branch-free)
    const float* xb = a.x + (int64_t)b * a.H * a.W * a.Cin;
    float4 areg[MRG][KC];
#pragma unroll
    for (int i = 0; i < MRG; ++i) {
        int rg = wave + 4 * i;
        if (rg > NRG - 1) rg = NRG - 1;
        const int m = rg * 16 + l15;
        const int ty_i = m / IW, tx_i = m - ty_i * IW;
        const int iy = iy0 + ty_i, ix = ix0 + tx_i;
        const int iyc = iy < 0 ? 0 : (iy > a.H - 1 ? a.H - 1 : iy), ixc = ix < 0 ? 0 : (ix > a.W - 1 ? a.W - 1 : ix);
        const float* px = xb + ((int64_t)iyc * a.W + ixc) * a.Cin;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            int koff = c * 16 + 4 * q;
            if (koff > a.Cin - 4) koff = a.Cin - 4;
            areg[i][c] = ld4(px + koff);
        }
    }
    // ---- the weights of a chunk live in REGISTERS: with the roles swapped (weights = MFMA operand A) a lane only ever needs
    // w[k = 16c + 4q + j][n = 16t + l15], i.e. KC*4*NT expand and NT*4*NTO project values per chunk.  No LDS staging, no barrier
    // for it; the next chunk's values are requested as soon as the current ones have been consumed.
    float wa[KC][4][NT];
    float wp[NT][4][NTO];
    auto load_wa = [&](int n0) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * c + 4 * q + j;
                const int kc = k < a.Cin ? k : a.Cin - 1;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float v = a.w_exp[(int64_t)kc * a.Cexp + n0 + 16 * t + l15];
                    wa[c][j][t] = k < a.Cin ? v : 0.f;
                }
            }
    };
    auto load_wp = [&](int n0) {
#pragma unroll
        for (int kc = 0; kc < NT; ++kc)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NTO; ++t) {
                    const int n = 16 * t + l15;
                    const int nc = n < a.Cout ? n : a.Cout - 1;
                    const float v = a.w_pj[(int64_t)(n0 + 16 * kc + 4 * q + j) * a.Cout + nc];
                    wp[kc][j][t] = n < a.Cout ? v : 0.f;
                }
    };
    load_wa(0);
    load_wp(0);
    for (int e = tid; e < 13 * a.Cexp; e += 256) {
        const int which = e / a.Cexp, c = e - which * a.Cexp;
        sVec[e] = which == 0 ? a.sc_e[c] : which == 1 ? a.sh_e[c] : which == 2 ? a.sc_d[c] : which == 3 ? a.sh_d[c] : a.w_dw[(which - 4) * a.Cexp + c];
    }

    f32x4 out[MRO][NTO];
#pragma unroll
    for (int i = 0; i < MRO; ++i)
#pragma unroll
        for (int t = 0; t < NTO; ++t) out[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ci = 0; ci < a.chunks; ++ci) {
        const int n0 = ci * CC;
        const int n_next = ci + 1 < a.chunks ? n0 + CC : n0;       // the last chunk re-requests itself: no branch around the loads
        __syncthreads();                               // sVec staged (first pass) / the previous chunk's phase 2 is done with sAct

        // ---- phase 1: expand GEMM over the input tile (halo included), BN + ReLU6, into LDS
#pragma unroll
        for (int i = 0; i < MRG; ++i) {
            const int rg = wave + 4 * i;
            if (rg < NRG) {                            // wave-uniform
                f32x4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const bool ok = c * 16 + 4 * q < a.Cin;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float xv = j == 0 ? areg[i][c].x : j == 1 ? areg[i][c].y : j == 2 ? areg[i][c].z : areg[i][c].w;
                        xv = ok ? xv : 0.f;
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j][t], xv, acc[t], 0, 0, 0);
                    }
                }
                const int m = rg * 16 + l15;
                const int ty_i = m / IW, tx_i = m - ty_i * IW;
                const int iy = iy0 + ty_i, ix = ix0 + tx_i;
                const bool inside = m < NPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int c4 = 16 * t + 4 * q;
                    const float4 sc = ld4(sVec + n0 + c4), sh = ld4(sVec + a.Cexp + n0 + c4);
                    float4 v;
                    v.x = inside ? apply_act(acc[t][0] * sc.x + sh.x, a.act_e) : 0.f;
                    v.y = inside ? apply_act(acc[t][1] * sc.y + sh.y, a.act_e) : 0.f;
                    v.z = inside ? apply_act(acc[t][2] * sc.z + sh.z, a.act_e) : 0.f;
                    v.w = inside ? apply_act(acc[t][3] * sc.w + sh.w, a.act_e) : 0.f;
                    st4(sAct + m * AP + c4, v);
                }
            }
        }
        load_wa(n_next);                               // in flight across the depthwise / project phases
        __syncthreads();

        // ---- phases 2 + 3 per wave and 16-pixel output row group: lane (l15, q) forms the depthwise result of pixel l15 for the
        // channels 16kc + 4q .. +3 — exactly its operand of the project MFMAs (k = 16kc + 4q + j), so `d` stays in registers
#pragma unroll
        for (int i = 0; i < MRO; ++i) {
            const int P = (wave + 4 * i) * 16 + l15;
            const int ly = P / TW, lx = P - ly * TW;
            const float* tap0 = sAct + ((ly * S) * IW + lx * S) * AP + 4 * q;
#pragma unroll
            for (int kc = 0; kc < NT; ++kc) {
                const float* wv = sVec + 4 * a.Cexp + n0 + 16 * kc + 4 * q;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const float4 v = ld4(tap0 + (ii * IW + jj) * AP + 16 * kc);
                        const float4 w4 = ld4(wv + (ii * 3 + jj) * a.Cexp);
                        acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                        acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
                    }
                const float4 sc = ld4(sVec + 2 * a.Cexp + n0 + 16 * kc + 4 * q), sh = ld4(sVec + 3 * a.Cexp + n0 + 16 * kc + 4 * q);
                float dv[4];
                dv[0] = apply_act(acc.x * sc.x + sh.x, a.act_d); dv[1] = apply_act(acc.y * sc.y + sh.y, a.act_d);
                dv[2] = apply_act(acc.z * sc.z + sh.z, a.act_d); dv[3] = apply_act(acc.w * sc.w + sh.w, a.act_d);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NTO; ++t) out[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[kc][j][t], dv[j], out[i][t], 0, 0, 0);
            }
        }
        load_wp(n_next);                               // in flight across the next chunk's expand phase
    }

    // ---- project epilogue: BN, residual = block input at the output pixel (stride 1, Cin == Cout), 16-byte stores
    float* yb = a.y + (int64_t)b * a.Ho * a.Wo * a.Cout;
#pragma unroll
    for (int i = 0; i < MRO; ++i) {
        const int P = (wave + 4 * i) * 16 + l15;
        const int ly = P / TW, lx = P - ly * TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        if (oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            const int c4 = 16 * t + 4 * q;
            if (c4 >= a.Cout) continue;
            const float4 sc = ld4(a.sc_p + c4), sh = ld4(a.sh_p + c4);
            float4 v;
            v.x = apply_act(out[i][t][0] * sc.x + sh.x, a.act_p); v.y = apply_act(out[i][t][1] * sc.y + sh.y, a.act_p);
            v.z = apply_act(out[i][t][2] * sc.z + sh.z, a.act_p); v.w = apply_act(out[i][t][3] * sc.w + sh.w, a.act_p);
            if (a.residual) {
                const float4 r = ld4(xb + ((int64_t)oy * a.W + ox) * a.Cin + c4);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            st4(yb + ((int64_t)oy * a.Wo + ox) * a.Cout + c4, v);
        }
    }
}

template <int S, int NT, int TH, int TW, int KC, int NTO>
static int launch_blk_k(BlkArgs a, hipStream_t st) {
    constexpr int CC = 16 * NT;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int NRG = (IH * IW + 15) / 16;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = cdiv(a.Ho, TH);
    a.chunks = a.Cexp / CC;
    const size_t lds = ((size_t)13 * a.Cexp + (size_t)NRG * 16 * (CC + 4)) * sizeof(float);
    AMS_REQUIRE(lds <= 150 * 1024, "block kernel: tile needs %zu bytes of LDS", lds);
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        AMS_CHECK_HIP(hipFuncSetAttribute((const void*)block_kernel<S, NT, TH, TW, KC, NTO>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    const int64_t nblocks = (int64_t)a.tiles_x * a.tiles_y * a.B;
    AMS_REQUIRE(nblocks > 0 && nblocks < 0x7fffffffLL, "block kernel: bad grid");
    static const std::string nm = "block_kernel<" + std::to_string(S) + ", " + std::to_string(NT) + ", " + std::to_string(TH) + ", " +
                                  std::to_string(TW) + ", " + std::to_string(KC) + ", " + std::to_string(NTO) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((block_kernel<S, NT, TH, TW, KC, NTO>), dim3((unsigned)nblocks), dim3(256), lds, st, a, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int S, int NT, int TH, int TW, int KC>
static int launch_blk_o(const BlkArgs& a, hipStream_t st) {
    return a.Cout <= 32 ? launch_blk_k<S, NT, TH, TW, KC, 2>(a, st) : launch_blk_k<S, NT, TH, TW, KC, 4>(a, st);
}

template <int S, int NT, int TH, int TW>
static int launch_blk_t(const BlkArgs& a, hipStream_t st) {
    return (a.Cin + 15) / 16 == 1 ? launch_blk_o<S, NT, TH, TW, 1>(a, st) : launch_blk_o<S, NT, TH, TW, 2>(a, st);
}

bool block_fused_supported(int Cin, int Cexp, int Cout, int stride, int rate, bool residual) {
    if (Cin % 4 != 0 || Cin > 32 || rate != 1 || (stride != 1 && stride != 2)) return false;
    if (Cout % 4 != 0 || Cout > 64) return false;
    if (residual && (stride != 1 || Cin != Cout)) return false;
    return Cexp % 32 == 0 || Cexp % 48 == 0;
}

int launch_block_fused(const float* x, int B, int H, int W, int Cin, const float* w_exp, const float* sc_e, const float* sh_e, int act_e, int Cexp,
                       const float* w_dw, int stride, const float* sc_d, const float* sh_d, int act_d, const float* w_pj, const float* sc_p,
                       const float* sh_p, int act_p, int Cout, bool residual, float* y, hipStream_t st) {
    AMS_REQUIRE(block_fused_supported(Cin, Cexp, Cout, stride, 1, residual), "block kernel: unsupported shape Cin=%d Cexp=%d Cout=%d s=%d", Cin, Cexp,
                Cout, stride);
    BlkArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.w_exp = w_exp; a.sc_e = sc_e; a.sh_e = sh_e; a.act_e = act_e; a.Cexp = Cexp;
    a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.w_pj = w_pj; a.sc_p = sc_p; a.sh_p = sh_p; a.act_p = act_p; a.Cout = Cout;
    a.residual = residual ? 1 : 0; a.y = y;
    same_pad(H, 3, stride, 1, &a.Ho, &a.pt);
    same_pad(W, 3, stride, 1, &a.Wo, &a.pl);
    const bool nt2 = Cexp % 32 == 0;
    // tile of output pixels per block: small enough that three or four blocks (12-16 waves) share a CU's LDS
    int th = 8, tw = 8;
    if (const char* e = getenv("AMS_BLK_TILE")) sscanf(e, "%dx%d", &th, &tw);          // tuning knob (tools/bench_block.py)
#define BLK(S_, NT_, TH_, TW_) if (stride == S_ && nt2 == (NT_ == 2) && th == TH_ && tw == TW_) return launch_blk_t<S_, NT_, TH_, TW_>(a, st);
    BLK(1, 2, 8, 16) BLK(1, 2, 16, 16) BLK(1, 2, 8, 8) BLK(1, 3, 8, 16) BLK(1, 3, 16, 16) BLK(1, 3, 8, 8)
    BLK(2, 2, 8, 8) BLK(2, 2, 8, 16) BLK(2, 3, 8, 8) BLK(2, 3, 8, 16)
#undef BLK
    set_error("block kernel: no %dx%d tile for stride %d", th, tw, stride);
    return AMS_E_INVALID;
}

}  // namespace ams

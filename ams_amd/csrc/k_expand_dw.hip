// Fused expand (1x1, BN, ReLU6) -> depthwise 3x3 (BN, ReLU6) of an inverted-residual block, frozen inference.
//
// In the layer-by-layer plan the 6x-expanded tensor is the dominant traffic: the expand GEMM writes it and the
// depthwise conv reads it back (2 x 40.3 M elements per 512x1024 frame = 46 % of all activation bytes).  Here one
// block produces a TH x TW tile of depthwise OUTPUT pixels for a chunk of 16*NT expanded channels:
//   phase 1  exact-f32 MFMA GEMM over the tile's input pixels incl. the 3x3 halo ((TH-1)*S + 2R + 1 rows/cols),
//            gathered straight from the block input (Cin = 16..64 channels), BN + ReLU6, result kept in LDS
//            (positions outside the image are the depthwise conv's SAME zero padding: stored as 0);
//   phase 2  depthwise 3x3 from LDS, BN + ReLU6, written as 64*NT-byte runs per pixel.
// The expanded activation never reaches HBM.  Costs: the halo is recomputed per tile (x1.27 GEMM work for 16x16 tiles,
// stride 1) and every channel chunk re-reads the small input tile from L2.
#include "pw_common.hpp"

namespace ams {

struct XdwArgs {
    const float* x;          // [B, H, W, Cin]
    int B, H, W, Cin;
    const float* w_exp;      // [Cin, Cexp]
    const float* sc_e; const float* sh_e;       // folded BN of the expand layer
    int Cexp;
    const float* w_dw;       // [9, Cexp]
    const float* sc_d; const float* sh_d;       // folded BN of the depthwise layer
    int act_e, act_d;
    float* y;                // [B, Ho, Wo, Cexp]
    int Ho, Wo, pt, pl;
    int tiles_x, tiles_y, chunks;
    int chunk_splits;        // blocks per tile: each walks chunks / chunk_splits channel chunks
    // fine-tune forward (STATS): the depthwise layer's BN statistics of the RAW result on the way out — partial rows [tiles][2][Cexp] of
    // sum(y - center), sum((y - center)^2), one row per tile (a block owns the columns of its chunks)
    const float* center; float* part;
};

template <int S, int R, int NT, int TH, int TW, int KC, bool STATS = false>
__global__ __launch_bounds__(256) void expand_dw_kernel(XdwArgs a, unsigned nblocks) {
    constexpr int CC = 16 * NT;                       // expanded channels per block
    constexpr int IH = (TH - 1) * S + 2 * R + 1, IW = (TW - 1) * S + 2 * R + 1;
    constexpr int NPIX = IH * IW;
    constexpr int NRG = (NPIX + 15) / 16;             // 16-pixel row groups of the input tile
    constexpr int WP = CC + 4;                        // pitch of the weight panel rows
    constexpr int AP = CC + 4;                        // pitch of the activation tile rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int Kpad = KC * 16;                     // Cin rounded up to the 16-wide k chunks
    constexpr int MRG = (NRG + 3) / 4;                // row groups per wave
    float* sW = smem;                                 // [Kpad][WP]
    float* sAff = sW + Kpad * WP;                     // sc_e, sh_e, sc_d, sh_d : 4 x CC
    float* sDw = sAff + 4 * CC;                       // [9][CC]
    float* sAct = sDw + 9 * CC;                       // [NRG*16][AP]

    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int cs = lb % a.chunk_splits;                // this block's share of the channel chunks
    unsigned t1 = lb / a.chunk_splits;
    const int tx = t1 % a.tiles_x; t1 /= a.tiles_x;
    const int ty = t1 % a.tiles_y;
    const int b = t1 / a.tiles_y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;

    // ---- every operand of this wave's row groups is requested up front (MRG*KC float4 per lane): the block is
    // latency-bound otherwise (one L2/HBM round trip per 8 MFMAs).  Branch-free: clamped coordinates, select.
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

    // ---- the block walks its share of the channel chunks of the tile (all of them when the launch has enough tiles to fill
    // the chip): the input fragments above are gathered once — they do not depend on the chunk, only the weights change —
    // instead of once per (tile, chunk) block.
    const int chunks_per_block = a.chunks / a.chunk_splits;
    // chunk parameters (expand weights, BN vectors, depthwise taps) go global -> registers -> LDS; the registers of chunk
    // ci + 1 are requested before the depthwise phase of chunk ci, so their L2 latency is off the critical path
    constexpr int NWV = (Kpad * (CC / 4) + 255) / 256, NDW = (9 * CC + 255) / 256;
    float4 wpre[NWV];
    float dpre[NDW];
    float apre[4];
    auto fetch_params = [&](int n0) {
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            int e = tid + 256 * u;
            if (e > Kpad * (CC / 4) - 1) e = Kpad * (CC / 4) - 1;
            const int kk = e / (CC / 4), c4 = (e - kk * (CC / 4)) * 4;
            const int kc = kk < a.Cin ? kk : a.Cin - 1;
            const float4 v = ld4(a.w_exp + (int64_t)kc * a.Cexp + n0 + c4);
            wpre[u] = kk < a.Cin ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NDW; ++u) {
            int e = tid + 256 * u;
            if (e > 9 * CC - 1) e = 9 * CC - 1;
            dpre[u] = a.w_dw[(e / CC) * a.Cexp + n0 + (e % CC)];
        }
        const int ec = tid < CC ? tid : CC - 1;
        apre[0] = a.sc_e[n0 + ec]; apre[1] = a.sh_e[n0 + ec]; apre[2] = a.sc_d[n0 + ec]; apre[3] = a.sh_d[n0 + ec];
    };
    auto store_params = [&]() {
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e = tid + 256 * u;
            if (e < Kpad * (CC / 4)) { const int kk = e / (CC / 4), c4 = (e - kk * (CC / 4)) * 4; st4(sW + kk * WP + c4, wpre[u]); }
        }
#pragma unroll
        for (int u = 0; u < NDW; ++u) {
            const int e = tid + 256 * u;
            if (e < 9 * CC) sDw[e] = dpre[u];
        }
        if (tid < CC) { sAff[tid] = apre[0]; sAff[CC + tid] = apre[1]; sAff[2 * CC + tid] = apre[2]; sAff[3 * CC + tid] = apre[3]; }
    };
    fetch_params(cs * chunks_per_block * CC);
    for (int ci = 0; ci < chunks_per_block; ++ci) {
    const int n0 = (cs * chunks_per_block + ci) * CC;
    if (ci > 0) __syncthreads();                       // phase 2 of the previous chunk still reads sDw / sAff / sAct
    store_params();
    __syncthreads();
    if (ci + 1 < chunks_per_block) fetch_params(n0 + CC);      // block-uniform

    // ---- phase 1: expand GEMM over the input tile (halo included), BN + ReLU6, into LDS
#pragma unroll
    for (int i = 0; i < MRG; ++i) {
        const int rg = wave + 4 * i;
        if (rg < NRG) {                                // wave-uniform
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const bool ok = c * 16 + 4 * q < a.Cin;      // k beyond Cin: the clamped load is ignored (W rows are 0 too)
                const float* sB = sW + (c * 16 + 4 * q) * WP + l15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xv = j == 0 ? areg[i][c].x : j == 1 ? areg[i][c].y : j == 2 ? areg[i][c].z : areg[i][c].w;
                    xv = ok ? xv : 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[j * WP + 16 * t], xv, acc[t], 0, 0, 0);
                }
            }
            // lane owns channels 16t + 4q .. +3 of tile pixel m; outside the image the depthwise conv sees zeros
            const int m = rg * 16 + l15;
            const int ty_i = m / IW, tx_i = m - ty_i * IW;
            const int iy = iy0 + ty_i, ix = ix0 + tx_i;
            const bool inside = m < NPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c4 = 16 * t + 4 * q;
                const float4 sc = ld4(sAff + c4), sh = ld4(sAff + CC + c4);
                float4 v;
                v.x = inside ? apply_act(acc[t][0] * sc.x + sh.x, a.act_e) : 0.f;
                v.y = inside ? apply_act(acc[t][1] * sc.y + sh.y, a.act_e) : 0.f;
                v.z = inside ? apply_act(acc[t][2] * sc.z + sh.z, a.act_e) : 0.f;
                v.w = inside ? apply_act(acc[t][3] * sc.w + sh.w, a.act_e) : 0.f;
                st4(sAct + m * AP + c4, v);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 from LDS; thread = (output pixel, 4 channels), channel groups along lanes
    constexpr int CG = CC / 4;
    float* yb = a.y + (int64_t)b * a.Ho * a.Wo * a.Cexp + n0;
    // STATS: a thread keeps ONE channel group (its sums stay in registers): threads 0 .. TPC*CG-1 walk the pixels TPC apart
    constexpr int TPC = 256 / CG;                      // threads per channel group
    float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f), st2 = st1, ctr = st1;
    if constexpr (STATS) { if (tid < TPC * CG && a.center) ctr = ld4(a.center + n0 + 4 * (tid % CG)); }
    for (int item = STATS ? (tid < TPC * CG ? tid : TH * TW * CG) : tid; item < TH * TW * CG; item += STATS ? TPC * CG : 256) {
        const int cg = item % CG, p = item / CG;
        const int ly = p / TW, lx = p - ly * TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const int c4 = cg * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 v = ld4(sAct + ((ly * S + i * R) * IW + lx * S + j * R) * AP + c4);
                const float4 w4 = ld4(sDw + (i * 3 + j) * CC + c4);
                acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
            }
        const float4 sc = ld4(sAff + 2 * CC + c4), sh = ld4(sAff + 3 * CC + c4);
        float4 o;
        o.x = apply_act(acc.x * sc.x + sh.x, a.act_d); o.y = apply_act(acc.y * sc.y + sh.y, a.act_d);
        o.z = apply_act(acc.z * sc.z + sh.z, a.act_d); o.w = apply_act(acc.w * sc.w + sh.w, a.act_d);
        st4(yb + ((int64_t)oy * a.Wo + ox) * a.Cexp + c4, o);
        if constexpr (STATS) {
            const float4 d = sub4_pk(o, ctr);
            st1 = add4_pk(st1, d);
            st2 = add4_pk(st2, mul4_pk(d, d));
        }
    }
    if constexpr (STATS) {
        // the block's sums of this chunk: thread (cg, k) -> LDS, then one thread per (sum, channel) adds the TPC entries in a fixed order
        __syncthreads();                               // phase 2 has finished reading sAct
        float* sRed = sAct;                            // [TPC][2][CC]
        if (tid < TPC * CG) {
            const int cg = tid % CG, k = tid / CG;
            st4(sRed + (k * 2 + 0) * CC + 4 * cg, st1);
            st4(sRed + (k * 2 + 1) * CC + 4 * cg, st2);
        }
        __syncthreads();
        if (tid < 2 * CC) {
            float v = 0.f;
            for (int k = 0; k < TPC; ++k) v += sRed[k * 2 * CC + tid];
            const int which = tid / CC, c = tid - which * CC;
            a.part[((int64_t)(lb / a.chunk_splits) * 2 + which) * a.Cexp + n0 + c] = v;
        }
    }
    }   // chunk
}

template <int S, int R, int NT, int TH, int TW, int KC, bool STATS = false>
static int launch_xdw_k(XdwArgs a, hipStream_t st) {
    constexpr int CC = 16 * NT;
    constexpr int IH = (TH - 1) * S + 2 * R + 1, IW = (TW - 1) * S + 2 * R + 1;
    constexpr int NRG = (IH * IW + 15) / 16;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = cdiv(a.Ho, TH);
    a.chunks = a.Cexp / CC;
    constexpr int Kpad = KC * 16;
    const size_t lds = ((size_t)Kpad * (CC + 4) + 4 * CC + 9 * CC + (size_t)NRG * 16 * (CC + 4)) * sizeof(float);
    AMS_REQUIRE(lds <= 150 * 1024, "expand_dw: tile needs %zu bytes of LDS", lds);
    static_assert(!STATS || (256 / (CC / 4)) * 2 * CC <= NRG * 16 * (CC + 4), "the statistics' LDS scratch must fit the activation tile");
    RUN_RC(func_allow_lds((const void*)expand_dw_kernel<S, R, NT, TH, TW, KC, STATS>, lds > 64 * 1024 ? 150 * 1024 : lds));
    // all chunks in one block when there are enough tiles to fill the chip (measured: from ~2000 tiles on, ~1000 with 4+ chunks), else one block
    // per (tile, chunk) for parallelism
    const int64_t tiles = (int64_t)a.tiles_x * a.tiles_y * a.B;
    a.chunk_splits = (tiles >= 2048 || (tiles >= 1024 && a.chunks >= 4)) ? 1 : a.chunks;
    const int64_t nblocks = (int64_t)a.tiles_x * a.tiles_y * a.B * a.chunk_splits;
    AMS_REQUIRE(nblocks > 0 && nblocks < 0x7fffffffLL, "expand_dw: bad grid");
    static const std::string nm = "expand_dw_kernel<" + std::to_string(S) + ", " + std::to_string(R) + ", " + std::to_string(NT) + ", " +
                                  std::to_string(TH) + ", " + std::to_string(TW) + ", " + std::to_string(KC) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((expand_dw_kernel<S, R, NT, TH, TW, KC, STATS>), dim3((unsigned)nblocks), dim3(256), lds, st, a, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int S, int R, int NT, int TH, int TW>
static int launch_xdw_t(XdwArgs a, hipStream_t st) {
    if (a.part) {
        switch ((a.Cin + 15) / 16) {
            case 1: return launch_xdw_k<S, R, NT, TH, TW, 1, true>(a, st);
            case 2: return launch_xdw_k<S, R, NT, TH, TW, 2, true>(a, st);
            case 3: return launch_xdw_k<S, R, NT, TH, TW, 3, true>(a, st);
            default: return launch_xdw_k<S, R, NT, TH, TW, 4, true>(a, st);
        }
    }
    switch ((a.Cin + 15) / 16) {
        case 1: return launch_xdw_k<S, R, NT, TH, TW, 1>(a, st);
        case 2: return launch_xdw_k<S, R, NT, TH, TW, 2>(a, st);
        case 3: return launch_xdw_k<S, R, NT, TH, TW, 3>(a, st);
        default: return launch_xdw_k<S, R, NT, TH, TW, 4>(a, st);
    }
}

bool expand_dw_supported(int Cin, int Cexp, int stride, int rate) {
    if (Cin % 4 != 0 || Cin > 64) return false;                // late blocks (Cin >= 96): the input tile re-read per chunk dominates
    if (rate != 1) return false;
    if (stride != 1 && stride != 2) return false;
    return Cexp % 32 == 0 || Cexp % 48 == 0;
}

// rows of the statistics form: one per (frame, tile)
static void xdw_tile(int Cexp, int stride, int* th, int* tw) {
    const bool nt2 = Cexp % 32 == 0;
    if (stride == 1) { *th = nt2 ? 16 : 8; *tw = 16; } else { *th = 8; *tw = 8; }
}
size_t expand_dw_stats_scratch(int B, int H, int W, int Cexp, int stride) {
    int th, tw, Ho, Wo, p;
    xdw_tile(Cexp, stride, &th, &tw);
    same_pad(H, 3, stride, 1, &Ho, &p);
    same_pad(W, 3, stride, 1, &Wo, &p);
    return (size_t)B * cdiv(Ho, th) * cdiv(Wo, tw) * 2 * (size_t)Cexp;
}

int launch_expand_dw(const float* x, int B, int H, int W, int Cin, const float* w_exp, const float* sc_e, const float* sh_e, int act_e,
                     int Cexp, const float* w_dw, int stride, int rate, const float* sc_d, const float* sh_d, int act_d, float* y,
                     hipStream_t st, const float* stats_center, float* stats_part, int* stats_rows) {
    AMS_REQUIRE(expand_dw_supported(Cin, Cexp, stride, rate), "expand_dw: unsupported shape Cin=%d Cexp=%d s=%d r=%d", Cin, Cexp, stride, rate);
    XdwArgs a;
    memset(&a, 0, sizeof(a));
    a.center = stats_center; a.part = stats_part;
    if (stats_part) {
        AMS_REQUIRE(stats_rows, "expand_dw: statistics need a row count output");
        int th, tw, Ho, Wo, pp;
        xdw_tile(Cexp, stride, &th, &tw);
        same_pad(H, 3, stride, rate, &Ho, &pp);
        same_pad(W, 3, stride, rate, &Wo, &pp);
        *stats_rows = B * cdiv(Ho, th) * cdiv(Wo, tw);
    }
    a.x = x; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.w_exp = w_exp; a.sc_e = sc_e; a.sh_e = sh_e; a.act_e = act_e;
    a.Cexp = Cexp; a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d; a.y = y;
    same_pad(H, 3, stride, rate, &a.Ho, &a.pt);
    same_pad(W, 3, stride, rate, &a.Wo, &a.pl);
    const bool nt2 = Cexp % 32 == 0;
    if (stride == 1) return nt2 ? launch_xdw_t<1, 1, 2, 16, 16>(a, st) : launch_xdw_t<1, 1, 3, 8, 16>(a, st);
    return nt2 ? launch_xdw_t<2, 1, 2, 8, 8>(a, st) : launch_xdw_t<2, 1, 3, 8, 8>(a, st);
}

}  // namespace ams

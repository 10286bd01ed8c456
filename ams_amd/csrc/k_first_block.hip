// First block of the backbone in one kernel (frozen inference): frame -> pad 127.5 -> x/127.5-1 -> stem 3x3 s2 (3->32) + BN
// + ReLU6 -> depthwise 3x3 (32) + BN + ReLU6 -> project 1x1 (32->16) + BN  (layers 1-3, MobilenetV2/Conv and
// MobilenetV2/expanded_conv).
//
// Layer by layer this is the largest traffic item of the network: the 32-channel half-resolution tensor (16.9 MB per
// 512x1024 frame) is written by the stem, read and written by the depthwise conv and read by the project GEMM — 67 MB per
// frame against 1.6 MB in and 8.4 MB out.  Here a block owns an 8 x 16 tile of output pixels:
//   phase 1  stem on the tile + 1-pixel halo (10 x 18 positions; zeros outside the feature map = the depthwise conv's
//            SAME padding): the 27-tap receptive field gathered from the uint8 frame as the MFMA operand (exact f32,
//            K = 27 -> 32), BN + ReLU6, result in LDS;
//   phase 2  depthwise 3x3 from LDS, 4 output rows per work unit through a sliding register window, tap weights in
//            registers, BN + ReLU6, result in LDS;
//   phase 3  project 32 -> 16 as 8 MFMAs per 16 pixels from LDS, BN, float4 stores (16 channels = 64 B per pixel, 16
//            consecutive pixels per lane group: full lines without a transpose).
// Arithmetic per element is the same as in the three separate kernels (same tap order, same MFMA chunking).
#include "pw_common.hpp"
#include "split_bf16.hpp"

namespace ams {

struct FirstBlockArgs {
    const void* frames; int B, H, W;        // [B,H,W,3] uint8 or float
    float ps;                               // pixel scale 1/127.5
    const float* w_stem;                    // [27][32]
    const float* sc_s; const float* sh_s;   // folded BN of the stem
    const float* w_dw;                      // [9][32]
    const float* sc_d; const float* sh_d;
    const float* w_pj;                      // [32][16]
    const float* sc_p; const float* sh_p;
    int act_s, act_d, act_p;
    float* y;                               // [B,Ho,Wo,16]
    int Ho, Wo, pt, pl;                     // stem output size and its SAME padding on the 127.5-padded frame
    int tiles_x, tiles_y;
    const uint16_t* w_parts;                // X6 form: stem weights as three bf16 parts [part][32 channels][32 k], parts w_plane apart
    int64_t w_plane;
    // H16 form (AMS_MATMUL_SPLIT_F16): stem and project products on two fp16 parts (hi | lo 2^11, split_bf16.hpp), 3 MFMAs of 16x16x32 each:
    // hs = the stem's panels [part][32 channels][32 k], hj = the project layer's [part][16 channels][32 k]
    const uint16_t* hs; int64_t hs_plane;
    const uint16_t* hj; int64_t hj_plane;
};

// X6 form: index into the normalisation table of the tap (0..255 the byte, 256 the 127.5 padding row / column, 257 = zero: outside)
__device__ __forceinline__ int fb_frame_index(const uint8_t* img, int H, int W, int iy, int ix, int ch) {
    const bool inside = iy >= 0 && ix >= 0 && iy <= H && ix <= W;
    const bool pad = iy >= H || ix >= W;
    const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const int ixc = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    const int raw = img[((int64_t)iyc * W + ixc) * 3 + ch];
    return inside ? (pad ? 256 : raw) : 257;
}

template <typename TIn>
__device__ __forceinline__ float fb_frame_value(const TIn* img, int H, int W, int iy, int ix, int ch, float ps) {
    // coordinates in the 127.5-padded (H+1) x (W+1) image; outside of it the stem's SAME zero padding (see k_conv.hip)
    const bool inside = iy >= 0 && ix >= 0 && iy <= H && ix <= W;
    const bool pad = iy >= H || ix >= W;
    const int iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    const int ixc = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    float raw = (float)img[((int64_t)iyc * W + ixc) * 3 + ch];
    raw = pad ? 127.5f : raw;
    const float v = __fsub_rn(__fmul_rn(raw, ps), 1.0f);
    return inside ? v : 0.f;
}

// X6 (uint8 frames only): the stem's products as six bf16 MFMAs on three-part splits, like the late layers and the other early blocks
// (96 instead of 256 matrix-pipe cycles per 16 positions x 16 channels).  A frame value has 256 possible inputs (+ the 127.5 padding
// and the zero outside the padded frame), so the three parts of x * ps - 1 come from a 258-entry LDS table built per block: a tap
// costs one byte load, one shift and one ds_read_b64 instead of convert + multiply + subtract + split.  Not bit-identical to the
// exact-f32 stem (f32-level: the dropped terms are <= 2^-24 relative).
// H16: the same with two fp16 parts: the table entry of a byte is ONE dword (hi | lo << 16: a tap = byte load + ds_read_b32 + a share of two
// v_perm), three MFMAs per 16 channels, and the project layer's products likewise (the depthwise result is in [0, 6]).
template <typename TIn, bool X6 = false, bool H16 = false>
__global__ __launch_bounds__(256) void first_block_kernel(FirstBlockArgs a, unsigned nblocks) {
    static_assert(!(X6 && H16), "one split form at a time");
    constexpr bool SPL = X6 || H16;                   // the lane's taps are k = 8q .. 8q + 7 (one 16x16x32 MFMA per product)
    constexpr bool TAB = SPL && sizeof(TIn) == 1;     // float frames: the same parts by splitting in registers (same bits, more VALU)
    constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    constexpr int NRG = (NPIX + 15) / 16;             // 12 row groups of stem positions
    constexpr int P = 36;                             // pitch of the 32-channel rows (stride 144 B: conflict-free b128 passes)
    constexpr int PW = 20;                            // pitch of the project weight rows
    __shared__ __attribute__((aligned(16))) float sW[28 * P];          // stem weights [k][n]; row 27 = zeros for the padding taps
    __shared__ __attribute__((aligned(16))) float sS[NPIX * P];        // stem output tile (three blocks per CU: 52.7 KB)
    __shared__ __attribute__((aligned(16))) float sD[TH * TW * P];     // depthwise output tile
    __shared__ __attribute__((aligned(16))) float sDw[9 * 32];
    __shared__ __attribute__((aligned(16))) float sWp[32 * PW];
    __shared__ __attribute__((aligned(16))) float sAff[32 * 4 + 16 * 2];      // sc_s, sh_s, sc_d, sh_d, sc_p, sh_p
    __shared__ __attribute__((aligned(8))) uint2 sTab[(TAB && X6) ? 258 : 1];   // X6: {hi | mid << 16, lo} bf16 parts of the normalised value
    __shared__ unsigned sTabH[(TAB && H16) ? 258 : 1];                          // H16: hi | lo << 16, fp16 parts

    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int tx = lb % a.tiles_x;
    unsigned t1 = lb / a.tiles_x;
    const int ty = t1 % a.tiles_y;
    const int b = t1 / a.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const float lo_s = a.act_s == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_s = a.act_s == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    const float lo_d = a.act_d == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_d = a.act_d == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();

    if constexpr (TAB) {
        for (int e = tid; e < 258; e += 256) {
            const float raw = e < 256 ? (float)e : 127.5f;
            const float val = e < 257 ? __fsub_rn(__fmul_rn(raw, a.ps), 1.0f) : 0.f;
            if constexpr (H16) {
                unsigned short h, l;
                split1_f16(val, h, l);
                sTabH[e] = (unsigned)h | ((unsigned)l << 16);
            } else {
                unsigned h, m, l;
                split_pair(val, 0.f, h, m, l);
                sTab[e] = make_uint2((h & 0xffffu) | (m << 16), l & 0xffffu);
            }
        }
    } else if constexpr (!SPL) {
        for (int e = tid; e < 28 * 32; e += 256) {
            const int kk = e >> 5, nn = e & 31;
            sW[kk * P + nn] = kk < 27 ? a.w_stem[kk * 32 + nn] : 0.f;
        }
    }
    for (int e = tid; e < 9 * 32; e += 256) sDw[e] = a.w_dw[e];
    if constexpr (!H16)
        for (int e = tid; e < 32 * 16; e += 256) sWp[(e >> 4) * PW + (e & 15)] = a.w_pj[e];
    if (tid < 32) {
        sAff[tid] = a.sc_s[tid]; sAff[32 + tid] = a.sh_s[tid]; sAff[64 + tid] = a.sc_d[tid]; sAff[96 + tid] = a.sh_d[tid];
    } else if (tid < 48) {
        sAff[128 + tid - 32] = a.sc_p[tid - 32]; sAff[144 + tid - 32] = a.sh_p[tid - 32];
    }

    // ---- phase 1: stem over the halo tile.  This lane's 8 taps: k = 16c + 4q + j -> (dy, dx, channel); k >= 27 padding
    int tdy[8], tdx[8], tch[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = SPL ? 8 * q + u : 16 * (u >> 2) + 4 * q + (u & 3);     // split forms: one 16x16x32 MFMA covers k = 8q .. 8q + 7
        const int tap = k / 3;
        tch[u] = k < 27 ? k - tap * 3 : -1;
        tdy[u] = tap / 3;
        tdx[u] = tap - tdy[u] * 3;
    }
    const TIn* img = reinterpret_cast<const TIn*>(a.frames) + (int64_t)b * a.H * a.W * 3;
    constexpr int MRG = (NRG + 3) / 4;
    float v[TAB ? 1 : MRG][8];
    int vi[TAB ? MRG : 1][8];                         // table form: table indices of the taps
    bool live[MRG];
    u32x4 wh[2][2];                                   // H16: the same as fp16 parts [t][hi | lo]
    if constexpr (H16) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
                wh[t][pp] = *reinterpret_cast<const u32x4*>(a.hs + pp * a.hs_plane + (int64_t)(16 * t + l15) * 32 + 8 * q);
    }
    bf16x8 wq[2][3];                                  // X6: this lane's stem-weight fragments (channel 16 t + l15, k = 8q .. 8q + 7), three parts
    if constexpr (X6) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
                wq[t][pp] = *reinterpret_cast<const bf16x8*>(a.w_parts + pp * a.w_plane + (int64_t)(16 * t + l15) * 32 + 8 * q);
    }
    // Interior tiles (all but the outermost ring): every stem position of the halo tile exists and every tap lies inside the
    // H x W frame, so a tap is one add, one byte load and the two-rounding normalisation — no clamps, no pad selects (the
    // border arithmetic was a quarter of the kernel's issue cycles).  Block-uniform branch.
    const bool interior = oy0 >= 1 && oy0 + TH < a.Ho && ox0 >= 1 && ox0 + TW < a.Wo &&
                          (oy0 - 1) * 2 - a.pt >= 0 && (oy0 + TH) * 2 - a.pt + 2 <= a.H - 1 &&
                          (ox0 - 1) * 2 - a.pl >= 0 && (ox0 + TW) * 2 - a.pl + 2 <= a.W - 1;
    if (interior) {
        int toff[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) toff[u] = tch[u] >= 0 ? (tdy[u] * a.W + tdx[u]) * 3 + tch[u] : 0;
#pragma unroll
        for (int i = 0; i < MRG; ++i) {
            int rg = wave + 4 * i;
            if (rg > NRG - 1) rg = NRG - 1;
            int m = rg * 16 + l15;
            live[i] = m < NPIX;
            if (m > NPIX - 1) m = NPIX - 1;
            const int ty_i = m / IW, tx_i = m - ty_i * IW;
            const int iy0 = (oy0 - 1 + ty_i) * 2 - a.pt, ix0 = (ox0 - 1 + tx_i) * 2 - a.pl;
            const TIn* p0 = img + ((int64_t)iy0 * a.W + ix0) * 3;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (TAB) {
                    const int raw = p0[toff[u]];
                    vi[i][u] = tch[u] >= 0 ? raw : 257;
                } else {
                    const float t = __fsub_rn(__fmul_rn((float)p0[toff[u]], a.ps), 1.0f);
                    v[i][u] = tch[u] >= 0 ? t : 0.f;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MRG; ++i) {
            int rg = wave + 4 * i;
            if (rg > NRG - 1) rg = NRG - 1;
            const int m = rg * 16 + l15;
            const int ty_i = m / IW, tx_i = m - ty_i * IW;
            const int sy = oy0 - 1 + ty_i, sx = ox0 - 1 + tx_i;              // position in the stem's output map
            live[i] = m < NPIX && sy >= 0 && sy < a.Ho && sx >= 0 && sx < a.Wo;
            const int iy0 = sy * 2 - a.pt, ix0 = sx * 2 - a.pl;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (TAB) {
                    const int idx = fb_frame_index(reinterpret_cast<const uint8_t*>(img), a.H, a.W, iy0 + tdy[u], ix0 + tdx[u], tch[u] < 0 ? 0 : tch[u]);
                    vi[i][u] = tch[u] >= 0 ? idx : 257;
                } else {
                    const float t = fb_frame_value(img, a.H, a.W, iy0 + tdy[u], ix0 + tdx[u], tch[u] < 0 ? 0 : tch[u], a.ps);
                    v[i][u] = tch[u] >= 0 ? t : 0.f;
                }
            }
        }
    }
    __syncthreads();                                                      // weights and coefficients are staged
#pragma unroll
    for (int i = 0; i < MRG; ++i) {
        const int rg = wave + 4 * i;
        if (rg < NRG) {                                                   // wave-uniform
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (H16) {
                unsigned e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (TAB) e[u] = sTabH[vi[i][u]];
                    else {
                        unsigned short h, l;
                        split1_f16(v[i][u], h, l);
                        e[u] = (unsigned)h | ((unsigned)l << 16);
                    }
                }
                u32x4 xh, xl;                                              // the lane's 8 taps: element u of the hi / lo part
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x05040100u);
                    xl[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x07060302u);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {                              // cross terms in their own accumulator, then the main term (k_pw_f16.hip)
                    f32x4 accx = (f32x4){0.f, 0.f, 0.f, 0.f};
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[t][1]), __builtin_bit_cast(f16x8, xh), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[t][0]), __builtin_bit_cast(f16x8, xl), accx, 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[t][0]), __builtin_bit_cast(f16x8, xh), acc[t], 0, 0, 0);
                    acc[t] = combine_f16(acc[t], accx);
                }
            } else if constexpr (X6) {
                uint2 e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (TAB) e[u] = sTab[vi[i][u]];
                    else {
                        unsigned h, m, l;
                        split_pair(v[i][u], 0.f, h, m, l);
                        e[u] = make_uint2((h & 0xffffu) | (m << 16), l & 0xffffu);
                    }
                }
                u32x4 p0, p1, p2;                                          // the lane's 8 taps as bf16 parts: element u of part p
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p0[j] = __builtin_amdgcn_perm(e[2 * j + 1].x, e[2 * j].x, 0x05040100u);
                    p1[j] = __builtin_amdgcn_perm(e[2 * j + 1].x, e[2 * j].x, 0x07060302u);
                    p2[j] = __builtin_amdgcn_perm(e[2 * j + 1].y, e[2 * j].y, 0x05040100u);
                }
                const bf16x8 x0 = __builtin_bit_cast(bf16x8, p0), x1 = __builtin_bit_cast(bf16x8, p1), x2 = __builtin_bit_cast(bf16x8, p2);
#pragma unroll
                for (int t = 0; t < 2; ++t) {                              // smallest terms first, as in the split GEMM
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][2], x0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][0], x2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][1], x1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][1], x0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][0], x1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t][0], x0, acc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 16 * (u >> 2) + 4 * q + (u & 3);
                const float* sB = sW + (k < 27 ? k : 27) * P + l15;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[0], v[i][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[16], v[i][u], acc[1], 0, 0, 0);
            }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n4 = 16 * t + 4 * q;
                const float4 sc = ld4(sAff + n4), sh = ld4(sAff + 32 + n4);
                // one v_med3 per value: clamp to the activation's bounds, both 0 outside the feature map (the depthwise conv's padding)
                const float lo = live[i] ? lo_s : 0.f, hi = live[i] ? hi_s : 0.f;
                float4 o;
                const float4 bn = muladd4_pk(make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]), sc, sh);
                o.x = __builtin_amdgcn_fmed3f(bn.x, lo, hi);
                o.y = __builtin_amdgcn_fmed3f(bn.y, lo, hi);
                o.z = __builtin_amdgcn_fmed3f(bn.z, lo, hi);
                o.w = __builtin_amdgcn_fmed3f(bn.w, lo, hi);
                if (rg * 16 + l15 < NPIX) st4(sS + (rg * 16 + l15) * P + n4, o);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 from LDS.  Unit = (4 channels, column, group of 4 rows): 256 units, one per thread
    {
        constexpr int RG = 4, WIN = RG + 2;
        const int cg = tid & 7, lx = (tid >> 3) & 15, g = tid >> 7;
        const int c4 = cg * 4;
        float4 w[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = ld4(sDw + k * 32 + c4);
        const float4 sc = ld4(sAff + 64 + c4), sh = ld4(sAff + 96 + c4);
        const float* col = sS + ((g * RG) * IW + lx) * P + c4;
        float4 acc[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < WIN; ++u) {
            float4 x3[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) x3[j] = ld4(col + (u * IW + j) * P);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (u - i >= 0 && u - i < RG) {                           // input row u is tap row i of output row u - i
                    float4& o = acc[u - i];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        fma4_pk(o, x3[j], w[i * 3 + j]);          // this phase is barrier-separated from the MFMA phases: packed f32 pays
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float4 o;
            const float4 bn = muladd4_pk(acc[r], sc, sh);
            o.x = __builtin_amdgcn_fmed3f(bn.x, lo_d, hi_d); o.y = __builtin_amdgcn_fmed3f(bn.y, lo_d, hi_d);
            o.z = __builtin_amdgcn_fmed3f(bn.z, lo_d, hi_d); o.w = __builtin_amdgcn_fmed3f(bn.w, lo_d, hi_d);
            st4(sD + ((g * RG + r) * TW + lx) * P + c4, o);
        }
    }
    __syncthreads();

    // ---- phase 3: project 32 -> 16.  Pixel group = one tile row (16 pixels); wave w takes rows 2w, 2w + 1
    float* yb = a.y + (int64_t)b * a.Ho * a.Wo * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * wave + i;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (H16) {                           // k = 8q .. 8q + 7 of pixel l15: one 16x16x32 MFMA per product
            const float* dp = sD + (row * TW + l15) * P + 8 * q;
            f16x8 dh, dl;
            split8_f16(ld4(dp), ld4(dp + 4), dh, dl);
            const f16x8 jh = *reinterpret_cast<const f16x8*>(a.hj + (int64_t)l15 * 32 + 8 * q);
            const f16x8 jl = *reinterpret_cast<const f16x8*>(a.hj + a.hj_plane + (int64_t)l15 * 32 + 8 * q);
            f32x4 accx = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(jl, dh, accx, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(jh, dl, accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(jh, dh, acc, 0, 0, 0);
            acc = combine_f16(acc, accx);
        } else
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 x4 = ld4(sD + (row * TW + l15) * P + 16 * c + 4 * q);
            const float* sB = sWp + (16 * c + 4 * q) * PW + l15;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[0], x4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[PW], x4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[2 * PW], x4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sB[3 * PW], x4.w, acc, 0, 0, 0);
        }
        const float4 sc = ld4(sAff + 128 + 4 * q), sh = ld4(sAff + 144 + 4 * q);
        float4 o;
        const float4 bn = muladd4_pk(make_float4(acc[0], acc[1], acc[2], acc[3]), sc, sh);
        o.x = apply_act(bn.x, a.act_p); o.y = apply_act(bn.y, a.act_p);
        o.z = apply_act(bn.z, a.act_p); o.w = apply_act(bn.w, a.act_p);
        const int oy = oy0 + row, ox = ox0 + l15;
        if (oy < a.Ho && ox < a.Wo) st4(yb + ((int64_t)oy * a.Wo + ox) * 16 + 4 * q, o);
    }
}

int launch_first_block(const void* frames, int dtype, int B, int H, int W, float pixel_scale, const float* w_stem,
                       const float* sc_s, const float* sh_s, int act_s, const float* w_dw, const float* sc_d, const float* sh_d,
                       int act_d, const float* w_pj, const float* sc_p, const float* sh_p, int act_p, float* y, hipStream_t st,
                       const uint16_t* w_parts, int64_t w_plane, const uint16_t* h_stem, int64_t h_stem_plane, const uint16_t* h_pj, int64_t h_pj_plane) {
    AMS_REQUIRE(dtype == AMS_DT_U8 || dtype == AMS_DT_F32, "first_block: frames must be uint8 or float32");
    FirstBlockArgs a;
    memset(&a, 0, sizeof(a));
    a.frames = frames; a.B = B; a.H = H; a.W = W; a.ps = pixel_scale;
    a.w_stem = w_stem; a.sc_s = sc_s; a.sh_s = sh_s; a.act_s = act_s;
    a.w_dw = w_dw; a.sc_d = sc_d; a.sh_d = sh_d; a.act_d = act_d;
    a.w_pj = w_pj; a.sc_p = sc_p; a.sh_p = sh_p; a.act_p = act_p; a.y = y;
    a.w_parts = w_parts; a.w_plane = w_plane;
    same_pad(H + 1, 3, 2, 1, &a.Ho, &a.pt);
    same_pad(W + 1, 3, 2, 1, &a.Wo, &a.pl);
    a.tiles_x = cdiv(a.Wo, 16);
    a.tiles_y = cdiv(a.Ho, 8);
    const int64_t nblocks = (int64_t)a.tiles_x * a.tiles_y * B;
    AMS_REQUIRE(nblocks > 0 && nblocks < 0x7fffffffLL, "first_block: bad grid");
    if (h_stem && h_pj) {                           // two fp16 parts in the stem and the project layer (takes precedence over w_parts)
        a.hs = h_stem; a.hs_plane = h_stem_plane; a.hj = h_pj; a.hj_plane = h_pj_plane;
        note_kernel(dtype == AMS_DT_U8 ? "first_block_kernel<unsigned char, false, true>" : "first_block_kernel<float, false, true>");
        if (dtype == AMS_DT_U8) hipLaunchKernelGGL((first_block_kernel<uint8_t, false, true>), dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
        else hipLaunchKernelGGL((first_block_kernel<float, false, true>), dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
        AMS_CHECK_LAUNCH();
        return AMS_OK;
    }
    if (w_parts) {                                  // three-part split products in the stem (see the kernel)
        note_kernel(dtype == AMS_DT_U8 ? "first_block_kernel<unsigned char, true>" : "first_block_kernel<float, true>");
        if (dtype == AMS_DT_U8) hipLaunchKernelGGL((first_block_kernel<uint8_t, true>), dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
        else hipLaunchKernelGGL((first_block_kernel<float, true>), dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
        AMS_CHECK_LAUNCH();
        return AMS_OK;
    }
    note_kernel(dtype == AMS_DT_U8 ? "first_block_kernel<unsigned char>" : "first_block_kernel<float>");
    if (dtype == AMS_DT_U8) hipLaunchKernelGGL(first_block_kernel<uint8_t>, dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
    else hipLaunchKernelGGL(first_block_kernel<float>, dim3((unsigned)nblocks), dim3(256), 0, st, a, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

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
    // the rectangle of interior tiles [ity0, ity1] x [itx0, itx1] (every stem position of the halo tile exists, every tap lies inside the frame);
    // nix * niy = 0: none.  border_only: first_block_kernel computes the tiles outside the rectangle (first_block_walk_kernel the ones inside)
    int ity0, itx0, niy, nix, border_only;
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
    int tx, ty, b;
    if (a.border_only) {                              // the tiles outside the interior rectangle: rows above, rows below, then the side columns
        const unsigned per_frame = (unsigned)(a.tiles_x * a.tiles_y - a.nix * a.niy);
        b = lb / per_frame;
        unsigned j = lb - (unsigned)b * per_frame;
        const unsigned above = (unsigned)(a.ity0 * a.tiles_x), below = (unsigned)((a.tiles_y - a.ity0 - a.niy) * a.tiles_x);
        if (j < above) { ty = j / a.tiles_x; tx = j - ty * a.tiles_x; }
        else if (j < above + below) { j -= above; ty = j / a.tiles_x; tx = j - ty * a.tiles_x; ty += a.ity0 + a.niy; }
        else {
            j -= above + below;
            const unsigned side = (unsigned)(a.tiles_x - a.nix);
            ty = j / side;
            tx = j - ty * side;
            ty += a.ity0;
            if (tx >= a.itx0) tx += a.nix;
        }
    } else {
        tx = lb % a.tiles_x;
        const unsigned t1 = lb / a.tiles_x;
        ty = t1 % a.tiles_y;
        b = t1 / a.tiles_y;
    }
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

// Walking form of first_block_kernel<uint8_t, false, true> for the INTERIOR tiles (FirstBlockArgs::ity0 ..; the border tiles — clamps, padding
// classes, 3-4x the gather work — are a second launch of the one-tile kernel with border_only: keeping their path out of this kernel keeps its
// registers and its s_waitcnt free of them): the same arithmetic per element (same bits), a block walks the tiles t = blockIdx.x,
// blockIdx.x + gridDim.x, ... of the interior rectangle
//   * everything that does not depend on the tile is staged once per block: the fp16 table, the depthwise taps and coefficients in LDS, the weight
//     fragments of the stem and the project layer and the stem's coefficients in registers;
//   * the taps of an interior tile are TWO 8-byte loads per stem position instead of eight byte loads: the lane's taps k = 8q .. 8q + 7 are the
//     last q bytes of the 9-byte window row q - 1 followed by the first 8 - q bytes of row q (RGB rows are contiguous), joined by two v_perm with
//     per-lane selectors; byte loads cost the texture-address unit a quad-cycle pass per lane each (the one-tile kernel: 96 µs of its 303);
//   * the bytes of the NEXT tile are requested as soon as this tile's table look-ups have consumed the registers: they arrive under the stem's
//     MFMAs and the depthwise + project phase.  This only works without a single spill inside the walk: hipcc follows a scratch reload with
//     s_waitcnt vmcnt(0), which waits for the prefetch as well (measured: 40 % of the wave's cycles);
//   * depthwise and project are ONE phase: lane = (pixel column, 8 channels), two output rows per wave, so the depthwise result of a lane IS its
//     operand of the project MFMAs (no second LDS tile, no second barrier), and the stem tile is double-buffered: one barrier per tile.
// Taps k >= 27 read a valid byte and meet the zero rows of the weight panels (launch_split_weights_f16 pads with zeros).
//
// tools/ only (AMS_FB_ABL=32): shader-clock cycles per wave summed over the launch — [0] tile decode, [1] stem phase, [2] wait at the barrier,
// [3] depthwise + project phase, [6] wave-tiles; read and cleared by ams_debug_phase_cycles
__device__ unsigned long long g_fb_cycles[8];

// ABL: measurement-only ablations (AMS_FB_ABL=<bits>, wrong results): 1 no result stores, 2 no depthwise arithmetic, 4 no byte loads / table
// look-ups, 8 no project MFMAs, 16 no stem MFMAs
// BORDER: the tiles OUTSIDE the interior rectangle (a second launch): the same two loads per position without clamps — an offset before the
// batch's first byte is out of the buffer's range and reads 0, one inside another row or frame reads bytes that the tap's class replaces — plus
// the class of every tap (inside / the 127.5 padding row or column / outside: zero) from the position's three row and three column classes
// (7 VALU instructions per tap against the one-tile kernel's clamped byte gather: 35 instead of 65 us for the 6048 border tiles of 32 frames), the
// position's own existence in the stem's output map, and bounds on the result stores.
template <int ABL = 0, bool BORDER = false>
__global__ __launch_bounds__(256, BORDER ? 2 : 3) void first_block_walk_kernel(FirstBlockArgs a, unsigned ntiles) {      // (BORDER: 2 blocks per CU — its extra state spills at 168 registers, and a spill reload waits for the prefetch)
    constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, NPIX = IH * IW, NRG = (NPIX + 15) / 16, MRG = 3, P = 36;
    static_assert(NRG == 4 * MRG, "three row groups of stem positions per wave");
    __shared__ __attribute__((aligned(16))) float sS[2][NPIX * P];     // stem tile, double-buffered: 3 blocks per CU = 3 x 54272 B
    __shared__ __attribute__((aligned(16))) float sDw[9 * 32];
    __shared__ __attribute__((aligned(16))) float sAffD[64];           // sc_d, sh_d
    __shared__ unsigned sTabH[256];                                    // hi | lo << 16 of byte * ps - 1
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float lo_s = a.act_s == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_s = a.act_s == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    const float lo_d = a.act_d == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi_d = a.act_d == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();

    {
        unsigned short h, l;
        split1_f16(__fsub_rn(__fmul_rn((float)tid, a.ps), 1.0f), h, l);
        sTabH[tid] = (unsigned)h | ((unsigned)l << 16);
    }
    for (int e = tid; e < 9 * 32; e += 256) sDw[e] = a.w_dw[e];
    if (tid < 32) { sAffD[tid] = a.sc_d[tid]; sAffD[32 + tid] = a.sh_d[tid]; }
    float4 sc_s[2], sh_s[2];                          // stem coefficients of this lane's output channels 16 t + 4q .. + 3
#pragma unroll
    for (int t = 0; t < 2; ++t) { sc_s[t] = ld4(a.sc_s + 16 * t + 4 * q); sh_s[t] = ld4(a.sh_s + 16 * t + 4 * q); }
    const float4 sc_p = ld4(a.sc_p + 4 * q), sh_p = ld4(a.sh_p + 4 * q);
    u32x4 wh[2][2];                                   // stem weights, channel 16 t + l15, k = 8q .. 8q + 7, [t][hi | lo]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
            wh[t][pp] = *reinterpret_cast<const u32x4*>(a.hs + pp * a.hs_plane + (int64_t)(16 * t + l15) * 32 + 8 * q);
    const f16x8 jh = *reinterpret_cast<const f16x8*>(a.hj + (int64_t)l15 * 32 + 8 * q);
    const f16x8 jl = *reinterpret_cast<const f16x8*>(a.hj + a.hj_plane + (int64_t)l15 * 32 + 8 * q);
    const uint8_t* frames = reinterpret_cast<const uint8_t*>(a.frames);
    const int frame_bytes = a.H * a.W * 3;            // (the launcher checks that the batch of frames is below 2 GB)
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frames), 0, a.B * frame_bytes, 0x00020000);
    // interior tiles: piece A = 4 bytes from byte 9 - q of window row q - 1 (its first q bytes are taps 8q .. 9q - 1), piece B = the first 8 bytes of
    // window row q (taps 9q ..); lanes that need no A (q = 0) / no B (q = 3) read the other piece's address
    const int row_bytes = a.W * 3;
    const unsigned off_b = q < 3 ? (unsigned)(q * row_bytes) : (unsigned)(2 * row_bytes + 6);
    const unsigned off_a = q > 0 ? (unsigned)((q - 1) * row_bytes + 9 - q) : off_b;
    // byte i of the joined 8 bytes = i < q ? A[i] : B[i - q]; v_perm selector bytes: 0-3 = second operand's bytes, 4-7 = first operand's
    unsigned sel0 = 0, sel1 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sel0 |= (unsigned)(i < q ? 4 + i : i - q) << (8 * i);          // perm(A.lo, B.lo)
        sel1 |= (unsigned)(4 + i - q) << (8 * i);                       // perm(B.hi, B.lo): B bytes 4 + i - q
    }
    // this lane's stem positions: row groups wave, wave + 4, wave + 8 of the 10 x 18 halo tile
    // (the last row group has 4 positions: its other 12 lanes repeat them — same window, same result, same LDS row — so that no store is
    // predicated and the three row groups are one straight-line block that hipcc can interleave)
    unsigned poff[MRG];                               // byte offset of the position's window from the tile's first byte
    int prow[MRG];                                    // the position's row of the LDS tile
    int ppos[BORDER ? MRG : 1];                       // BORDER: its (row, column) in the halo tile
#pragma unroll
    for (int i = 0; i < MRG; ++i) {
        const int m = (wave + 4 * i) * 16 + l15;
        const int mc = m < NPIX ? m : NPIX - 4 + (l15 & 3);
        const int ty = mc / IW, tx = mc - ty * IW;
        poff[i] = (unsigned)(2 * ty * row_bytes + 2 * tx * 3);
        prow[i] = mc * P;
        if constexpr (BORDER) ppos[i] = ty | (tx << 8);
    }
    // BORDER: bit offsets of this lane's taps in the packed row / column classes (2 bits per dy / dx; offset 6 = the constant "outside" for k >= 27)
    unsigned tsel[2] = {0, 0};                        // six bits per tap (row offset | column offset << 3), four taps per register
    unsigned e_pad = 0;
    if constexpr (BORDER) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 8 * q + u, tap = k / 3, dy = tap / 3, dx = tap - dy * 3;
            tsel[u >> 2] |= (k < 27 ? (unsigned)(2 * dy) | ((unsigned)(2 * dx) << 3) : 6u | (6u << 3)) << (6 * (u & 3));
        }
        unsigned short h, l;
        split1_f16(__fsub_rn(__fmul_rn(127.5f, a.ps), 1.0f), h, l);
        e_pad = (unsigned)h | ((unsigned)l << 16);
    }

    struct Tile { int b, oy0, ox0; };
    auto decode = [&](unsigned t) {
        const unsigned lb = xcd_remap(t, ntiles);
        Tile r;
        if constexpr (BORDER) {                       // rows above the rectangle, rows below, then the side columns (as first_block_kernel's border_only)
            const unsigned per_frame = (unsigned)(a.tiles_x * a.tiles_y - a.nix * a.niy);
            r.b = (int)(lb / per_frame);
            unsigned j = lb - (unsigned)r.b * per_frame;
            const unsigned above = (unsigned)(a.ity0 * a.tiles_x), below = (unsigned)((a.tiles_y - a.ity0 - a.niy) * a.tiles_x);
            int ty, tx;
            if (j < above) { ty = (int)(j / a.tiles_x); tx = (int)j - ty * a.tiles_x; }
            else if (j < above + below) { j -= above; ty = (int)(j / a.tiles_x); tx = (int)j - ty * a.tiles_x; ty += a.ity0 + a.niy; }
            else {
                j -= above + below;
                const unsigned side = (unsigned)(a.tiles_x - a.nix);
                ty = (int)(j / side);
                tx = (int)j - ty * (int)side;
                ty += a.ity0;
                if (tx >= a.itx0) tx += a.nix;
            }
            r.oy0 = ty * TH; r.ox0 = tx * TW;
            return r;
        }
        const unsigned t1 = lb / a.nix;
        r.ox0 = (a.itx0 + (int)(lb - t1 * a.nix)) * TW;
        r.b = (int)(t1 / a.niy);
        r.oy0 = (a.ity0 + (int)(t1 - (unsigned)r.b * a.niy)) * TH;
        return r;
    };
    auto tile_base = [&](const Tile& tl) { return tl.b * frame_bytes + (((tl.oy0 - 1) * 2 - a.pt) * a.W + ((tl.ox0 - 1) * 2 - a.pl)) * 3; };
    int psh[BORDER ? MRG : 1];                        // BORDER: bit shifts that undo the clamps of the two loads (A's + 128 x B's)
    unsigned pa[MRG];                                 // the pieces of each row group, in flight or landed (A: at most its first 3 bytes are taps — a
    uint2 pb[MRG];                                    // dword; a loaded register that nothing reads would be reused, behind a wait for the load)
    auto request = [&](int base, int i) {
        unsigned p = poff[i];
        asm volatile("" : "+v"(p));                   // two adds per row group instead of six registers held over the walk
        if constexpr (BORDER) {
            // a piece may start before the batch's first byte or end behind its last one while some of its bytes are real taps (the first / last
            // pixels of the first / last frame): the load is moved inside and the bytes are shifted back (the vacated bytes belong to taps
            // whose class replaces them); shifts beyond a piece's length only occur where every tap is outside
            const int total = a.B * frame_bytes;
            const int oa = base + (int)(p + off_a), ob = base + (int)(p + off_b);
            const int ca = oa < 0 ? 0 : (oa > total - 4 ? total - 4 : oa), cb = ob < 0 ? 0 : (ob > total - 8 ? total - 8 : ob);
            pa[i] = __builtin_amdgcn_raw_buffer_load_b32(frsrc, ca, 0, 0);
            pb[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(frsrc, cb, 0, 0));
            psh[i] = (oa - ca < -3 ? -3 : (oa - ca > 3 ? 3 : oa - ca)) * 8 + 128 * ((ob - cb < -7 ? -7 : (ob - cb > 7 ? 7 : ob - cb)) * 8);
        } else {
            pa[i] = __builtin_amdgcn_raw_buffer_load_b32(frsrc, (int)(p + off_a), base, 0);
            pb[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(frsrc, (int)(p + off_b), base, 0));
        }
    };
    constexpr bool TIMED = (ABL & 32) != 0;
    unsigned long long tc[4] = {0, 0, 0, 0}, tl_ = 0, ntile = 0;
    auto lap = [&](int slot) {
        if constexpr (TIMED) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tc[slot] += now - tl_;
            tl_ = now;
        }
    };
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;                          // block-uniform
    Tile cur = decode(t);
    bool ready = false;                               // the pieces of `cur` are in flight / in registers (requested during the previous tile)
    int buf = 0;
    __syncthreads();                                  // table, taps and coefficients are staged
    if constexpr (TIMED) tl_ = __builtin_amdgcn_s_memtime();
    for (;;) {
        if (!ready && !(ABL & 4)) {                    // (the block's first tile)
            const int base = tile_base(cur);
#pragma unroll
            for (int i = 0; i < MRG; ++i) request(base, i);
        }
        const unsigned tn = t + gridDim.x;
        const bool more = tn < ntiles;                // block-uniform
        Tile nxt = cur;
        if (more) nxt = decode(tn);
        const int nbase = tile_base(nxt);
        float* sT = sS[buf];
        lap(0);
        // ---- phase 1: stem over the halo tile; the registers of a row group's pieces are refilled for the next tile as soon as the table
        // look-ups have consumed them
#pragma unroll
        for (int i = 0; i < MRG; ++i) {
            unsigned e[8];
            if constexpr ((ABL & 4) != 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = 0x3c00u + (unsigned)(lane + u + i);
            } else {
                unsigned va = pa[i];
                uint2 vb = pb[i];
                if constexpr (BORDER) {
                    const int sb = psh[i] >= 0 ? (psh[i] + 64) / 128 : -((64 - psh[i]) / 128);      // B's shift (a multiple of 8 in -56 .. 56) ...
                    const int sa = psh[i] - 128 * sb;                                                // ... and A's (-24 .. 24)
                    va = sa >= 0 ? va >> sa : va << -sa;
                    const unsigned long long w = ((unsigned long long)vb.y << 32) | vb.x;
                    const unsigned long long ws = sb >= 0 ? w >> sb : w << -sb;
                    vb = make_uint2((unsigned)ws, (unsigned)(ws >> 32));
                }
                const unsigned d0 = __builtin_amdgcn_perm(va, vb.x, sel0), d1 = __builtin_amdgcn_perm(vb.y, vb.x, sel1);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned d = u < 4 ? d0 : d1;
                    e[u] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(sTabH) + (((d >> (8 * (u & 3))) & 0xffu) << 2));
                }
            }
            if constexpr (!(ABL & 4)) request(nbase, i);        // unconditional (the last tile re-reads its own window): behind a branch hipcc's wait for the
                                                                // older pieces becomes vmcnt(0), which waits for these loads as well
            bool lives = true;
            if constexpr (BORDER) {
                const int sy = cur.oy0 - 1 + (ppos[i] & 255), sx = cur.ox0 - 1 + (ppos[i] >> 8);      // position in the stem's output map
                lives = sy >= 0 && sy < a.Ho && sx >= 0 && sx < a.Wo;
                const int iy0 = 2 * sy - a.pt, ix0 = 2 * sx - a.pl;
                unsigned rcp = 2u << 6, ccp = 2u << 6;       // classes of the window's rows / columns: 0 inside, 1 the 127.5 padding, 2 outside
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    const int iy = iy0 + dd, ix = ix0 + dd;
                    rcp |= (unsigned)(iy < 0 ? 2 : (iy < a.H ? 0 : (iy == a.H ? 1 : 2))) << (2 * dd);
                    ccp |= (unsigned)(ix < 0 ? 2 : (ix < a.W ? 0 : (ix == a.W ? 1 : 2))) << (2 * dd);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned sd = (tsel[u >> 2] >> (6 * (u & 3))) & 63u;
                    const unsigned rc = (rcp >> (sd & 7u)) & 3u, cc = (ccp >> (sd >> 3)) & 3u;
                    const unsigned cls = rc > cc ? rc : cc;
                    e[u] = cls == 0u ? e[u] : (cls == 1u ? e_pad : 0u);
                }
            }
            u32x4 xh, xl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x05040100u);
                xl[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x07060302u);
            }
            const float lo = lives ? lo_s : 0.f, hi = lives ? hi_s : 0.f;      // (interior tiles: always the activation's bounds; outside the map both 0: the depthwise conv's padding)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accx = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr ((ABL & 16) != 0) { acc[0] = __uint_as_float(xh[0] ^ wh[tt][0][1]); acc[1] = __uint_as_float(xl[1]); acc[2] = __uint_as_float(xh[2]); acc[3] = __uint_as_float(xl[3] ^ xh[3]); }
                else {
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[tt][1]), __builtin_bit_cast(f16x8, xh), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[tt][0]), __builtin_bit_cast(f16x8, xl), accx, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[tt][0]), __builtin_bit_cast(f16x8, xh), acc, 0, 0, 0);
                    acc = combine_f16(acc, accx);
                }
                const float4 bn = muladd4_pk(make_float4(acc[0], acc[1], acc[2], acc[3]), sc_s[tt], sh_s[tt]);
                float4 o;
                o.x = __builtin_amdgcn_fmed3f(bn.x, lo, hi);
                o.y = __builtin_amdgcn_fmed3f(bn.y, lo, hi);
                o.z = __builtin_amdgcn_fmed3f(bn.z, lo, hi);
                o.w = __builtin_amdgcn_fmed3f(bn.w, lo, hi);
                st4(sT + prow[i] + 16 * tt + 4 * q, o);
            }
        }
        lap(1);
        __syncthreads();
        lap(2);
        // ---- phase 2: depthwise 3x3 + project for output rows 2 wave, 2 wave + 1: lane = (pixel column l15, channels 8q .. 8q + 7), the depthwise
        // taps in the one-tile kernel's order (tap rows outer, columns inner, fma), four channels at a time
        {
            float4 d[2][2];                           // [row][channel half]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __builtin_amdgcn_sched_barrier(0);    // one half's taps and window in registers at a time (hipcc otherwise hoists every LDS read and spills)
                const int c4 = 8 * q + 4 * h;
                float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
                if constexpr (!(ABL & 2)) {
                    float4 w[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) w[k] = ld4(sDw + k * 32 + c4);
                    const float* col = sT + ((2 * wave) * IW + l15) * P + c4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (u == 2) __builtin_amdgcn_sched_barrier(0);
                        float4 x3[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) x3[j] = ld4(col + (u * IW + j) * P);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            if (u - i >= 0 && u - i < 2) {            // input row u is tap row i of output row u - i
#pragma unroll
                                for (int j = 0; j < 3; ++j) fma4_pk(acc[u - i], x3[j], w[i * 3 + j]);
                            }
                        }
                    }
                } else { acc[0] = ld4(sT + (2 * wave * IW + l15) * P + c4); acc[1] = acc[0]; }
                const float4 sc = ld4(sAffD + c4), sh = ld4(sAffD + 32 + c4);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float4 bn = muladd4_pk(acc[r], sc, sh);
                    d[r][h].x = __builtin_amdgcn_fmed3f(bn.x, lo_d, hi_d); d[r][h].y = __builtin_amdgcn_fmed3f(bn.y, lo_d, hi_d);
                    d[r][h].z = __builtin_amdgcn_fmed3f(bn.z, lo_d, hi_d); d[r][h].w = __builtin_amdgcn_fmed3f(bn.w, lo_d, hi_d);
                }
                // both rows of this half exist HERE: without the pin hipcc sinks the second row's arithmetic behind the first row's project MFMAs and
                // keeps (spills) both halves' taps and windows until then
                asm volatile("" : "+v"(d[0][h].x), "+v"(d[0][h].y), "+v"(d[0][h].z), "+v"(d[0][h].w), "+v"(d[1][h].x), "+v"(d[1][h].y), "+v"(d[1][h].z), "+v"(d[1][h].w));
            }
            __builtin_amdgcn_sched_barrier(0);
            float* yb = a.y + (int64_t)cur.b * a.Ho * a.Wo * 16;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f16x8 dh, dl;
                split8_f16(d[r][0], d[r][1], dh, dl);
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accx = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr ((ABL & 8) != 0) { acc[0] = d[r][0].x; acc[1] = d[r][0].y; acc[2] = d[r][1].z; acc[3] = d[r][1].w; }
                else {
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(jl, dh, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(jh, dl, accx, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(jh, dh, acc, 0, 0, 0);
                    acc = combine_f16(acc, accx);
                }
                const float4 bn = muladd4_pk(make_float4(acc[0], acc[1], acc[2], acc[3]), sc_p, sh_p);
                float4 o;
                o.x = apply_act(bn.x, a.act_p); o.y = apply_act(bn.y, a.act_p);
                o.z = apply_act(bn.z, a.act_p); o.w = apply_act(bn.w, a.act_p);
                const int oy = cur.oy0 + 2 * wave + r, ox = cur.ox0 + l15;
                if ((!BORDER || (oy < a.Ho && ox < a.Wo)) && (!(ABL & 1) || o.x == 12345.f)) st4(yb + ((int64_t)oy * a.Wo + ox) * 16 + 4 * q, o);
            }
        }
        lap(3);
        if constexpr (TIMED) ntile += 1;
        if (!more) break;
        t = tn;
        cur = nxt;
        ready = true;
        buf ^= 1;
    }
    if constexpr (TIMED) {
        if (lane == 0) {
            for (int i = 0; i < 4; ++i) atomicAdd(&g_fb_cycles[i], tc[i]);
            atomicAdd(&g_fb_cycles[6], ntile);
        }
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
        if (dtype == AMS_DT_U8 && knobs().fb_walk != 0 && (int64_t)B * H * W * 3 + 16 < 0x7fffffffLL) {
            // interior tiles (first_block_walk_kernel's header) form a rectangle: the conditions are separate in y and x
            auto range = [](int tiles, int T, int Osize, int pad, int Isize, int* first, int* count) {
                *first = 0; *count = 0;
                for (int t = 0; t < tiles; ++t) {
                    const int o0 = t * T;
                    const bool in = o0 >= 1 && o0 + T < Osize && (o0 - 1) * 2 - pad >= 0 && (o0 + T) * 2 - pad + 2 <= Isize - 1;
                    if (in) { if (!*count) *first = t; ++*count; }
                }
            };
            range(a.tiles_y, 8, a.Ho, a.pt, H, &a.ity0, &a.niy);
            range(a.tiles_x, 16, a.Wo, a.pl, W, &a.itx0, &a.nix);
            const int64_t n_in = (int64_t)B * a.niy * a.nix;
            if (n_in > 0) {
                int per_cu = 1, cus = 256;
                RUN_RC(func_blocks_per_cu((const void*)first_block_walk_kernel<0>, 256, 0, &per_cu));
                RUN_RC(device_cus(&cus));
                const int64_t slots = (int64_t)per_cu * cus;
                int64_t rounds = cdiv(n_in, slots);
                if (knobs().fb_walk > 0 && rounds > knobs().fb_walk) rounds = knobs().fb_walk;      // AMS_FB_WALK=<n>: at most n tiles per block
                int64_t grid = cdiv(cdiv(n_in, rounds), 8) * 8;           // equal shares; a multiple of 8: a block's tiles stay on its XCD's share
                if (grid > n_in) grid = n_in;
                note_kernel("first_block_walk_kernel<0>");                // (a profiled launch bracket covers the border launch below as well)
#ifdef AMS_MEASURE
                switch (knobs().fb_abl) {           // MEASUREMENT BUILD ONLY (libams_hip_measure.so: tools/sweep_fb_abl.sh, tools/fb_phases.py)
#define FB_A(A_) case A_: hipLaunchKernelGGL(first_block_walk_kernel<A_>, dim3((unsigned)grid), dim3(256), 0, st, a, (unsigned)n_in); break;
                    FB_A(1) FB_A(2) FB_A(4) FB_A(8) FB_A(16) FB_A(6) FB_A(14) FB_A(30) FB_A(31) FB_A(32) FB_A(33) FB_A(34) FB_A(36) FB_A(40) FB_A(48) FB_A(63)
#undef FB_A
                    default: hipLaunchKernelGGL(first_block_walk_kernel<0>, dim3((unsigned)grid), dim3(256), 0, st, a, (unsigned)n_in);
                }
#else
                hipLaunchKernelGGL(first_block_walk_kernel<0>, dim3((unsigned)grid), dim3(256), 0, st, a, (unsigned)n_in);
#endif
                AMS_CHECK_LAUNCH();
                const int64_t n_border = nblocks - n_in;
                if (n_border > 0) {
                    a.border_only = 1;
                    if (knobs().fb_walk == -2)          // AMS_FB_WALK=-2: the border tiles on the one-tile kernel (A/B of the two border forms)
                        hipLaunchKernelGGL((first_block_kernel<uint8_t, false, true>), dim3((unsigned)n_border), dim3(256), 0, st, a, (unsigned)n_border);
                    else {
                        int bper_cu = 1;
                        RUN_RC(func_blocks_per_cu((const void*)first_block_walk_kernel<0, true>, 256, 0, &bper_cu));
                        const int64_t bslots = (int64_t)bper_cu * cus;
                        int64_t brounds = cdiv(n_border, bslots);
                        if (knobs().fb_walk > 0 && brounds > knobs().fb_walk) brounds = knobs().fb_walk;
                        int64_t bgrid = cdiv(cdiv(n_border, brounds), 8) * 8;
                        if (bgrid > n_border) bgrid = n_border;
                        hipLaunchKernelGGL((first_block_walk_kernel<0, true>), dim3((unsigned)bgrid), dim3(256), 0, st, a, (unsigned)n_border);
                    }
                    AMS_CHECK_LAUNCH();
                }
                return AMS_OK;
            }
        }
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

// tools/ only: per-phase shader-clock sums, read and cleared: which = 0 the walking first block (AMS_FB_ABL=32), 1 block_kernel (AMS_BLK_TIMED=1), 2 xdw_wreg_kernel, 3 xdw_stream_kernel (AMS_XWR_TIMED=1)
extern "C" int ams_debug_phase_cycles(int32_t which, uint64_t* out, int32_t n) {
    unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (which == 1) { RUN_RC(ams::blk_phase_cycles(h)); }
    else if (which == 2) { RUN_RC(ams::xwr_phase_cycles(h)); }
    else if (which == 3) { RUN_RC(ams::xds_phase_cycles(h)); }
    else {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ams::g_fb_cycles), sizeof(h)) != hipSuccess) return AMS_E_HIP;
        if (hipMemcpyToSymbol(HIP_SYMBOL(ams::g_fb_cycles), z, sizeof(z)) != hipSuccess) return AMS_E_HIP;
    }
    for (int i = 0; i < n && i < 8; ++i) out[i] = h[i];
    return AMS_OK;
}

// Fine-tune step, stride-16 blocks: the depthwise layer's BN-backward apply pass (dz_d = A dy + B + C z_d, three tensor passes of up to
// 66 MB each at 8 x 512x1024) folded into the one-kernel depthwise backward of k_conv.hip (dw3x3_dgrad_bn_kernel).
//
// That kernel reads its nine taps from global memory, three columns per thread: applying (A, B, C) on those loads costs two loads and
// five VALU per tap, three times redundant along a row — measured no faster than the separate pass.  Here the
// taps go through LDS instead, so every dz_d value is formed ONCE:
//   * a block owns one image, one parity sub-image (rate 2 = four independent rate-1 problems on the (row, column) parity classes), a
//     strip of <= 16 sub-image columns, a 64-channel chunk (256 contiguous bytes per pixel) and a band of rows; thread = (column, 4 channels);
//   * walking down the rows it stages row t of dz_d = (A dy + B) + C z_d — the unfused operations of bn_bwd_apply_kernel, so the values
//     are bit-identical to the pass it replaces — into a four-row LDS ring (one halo column on each side is staged by 32 extra duties),
//     one barrier, then forms output row t - 1 from the ring: da_e = dwconv^T(dz_d) with the same tap order and fmaf chain as
//     dw3x3_dgrad_bn_kernel (the result is bit-identical), x act'(z_e sc + sh), and the BN-backward sums of the expand layer and the nine
//     taps of the depthwise weight gradient accumulate in registers as there;
//   * the loads of row t + 1 (dy, z_d) and of z_e for row t are in flight during the arithmetic of row t - 1.
// One partial row [11][C] per (image, sub-image, band, column strip), each block writing the 64-channel slice it owns; threads that share
// a channel group are folded in a fixed order through LDS (deterministic).
#include "kernels.hpp"

namespace ams {

struct Dg2Args {
    const float* dy; const float* zd; const float* cA; const float* cB; const float* cC;      // depthwise layer: masked gradient, raw output, BN-backward coefficients
    const float* w;                                                                           // depthwise weights [9][C]
    const float* ze; const float* sc; const float* sh; const float* mu; const float* rs; int act;     // expand layer: raw output, BN scale / shift / mean / rstd
    float* out; float* part;
    int B, H, W, C, R;
    int tiles_x, tw, bands, rows_per_band, cchunks;
};

constexpr int kDg2Cols = 16, kDg2Cg = 16;          // columns and channel groups (x 4 channels) of a block

template <int R>
__global__ __launch_bounds__(256) void dw3x3_dgrad_bn2_kernel(Dg2Args a) {
    constexpr int LC = kDg2Cols + 2;                // ring columns incl. the two halo columns
    constexpr int NQ = 8 + 36;                      // s1, s2, nine taps: 4 channels each
    // the ring (4 rows x 18 columns x 16 channel groups of float4 = 18 KB) and, after the walk, the block reduction's 256 x 45 floats
    __shared__ __attribute__((aligned(16))) float smem[256 * (NQ + 1)];
    float4* ring = reinterpret_cast<float4*>(smem);

    int bid = blockIdx.x;
    const int cc = bid % a.cchunks; bid /= a.cchunks;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int band = bid % a.bands; bid /= a.bands;
    const int sub = bid % (R * R);
    const int b = bid / (R * R);
    const int py = sub / R, px = sub - py * R;
    const int Hs = (a.H - py + R - 1) / R, Ws = (a.W - px + R - 1) / R;
    const int tid = threadIdx.x, tcol = tid >> 4, tcg = tid & 15;
    const int c0 = cc * (4 * kDg2Cg) + 4 * tcg;
    const bool chan_ok = c0 < a.C;
    const int c0c = chan_ok ? c0 : 0;
    const int j = tx * a.tw + tcol;                                    // this thread's sub-image column
    const bool col_ok = tcol < a.tw && j < Ws && chan_ok;
    const int jc = j < Ws ? j : Ws - 1;
    // halo duty (threads 0 .. 31): column tx * tw - 1 (side 0) or tx * tw + tw (side 1), ring column 0 or tw + 1
    const bool halo = tid < 32;
    const int hside = tid >> 4;
    const int hj = hside ? tx * a.tw + a.tw : tx * a.tw - 1;
    const bool hcol_ok = halo && hj >= 0 && hj < Ws && chan_ok;
    const int hjc = hj < 0 ? 0 : (hj < Ws ? hj : Ws - 1);
    const int hlc = hside ? a.tw + 1 : 0;

    const int i0 = band * a.rows_per_band;
    int i1 = i0 + a.rows_per_band;
    if (i1 > Hs) i1 = Hs;
    const int64_t img = (int64_t)b * a.H * a.W * a.C;
    auto pix = [&](int i, int jj) { return img + ((int64_t)(py + R * i) * a.W + (px + R * jj)) * a.C + c0c; };

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = zero4, s2 = zero4, dwv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv[k] = zero4;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(a.w + (8 - k) * a.C + c0c);          // the conv with the flipped kernel
    const float4 cA = ld4(a.cA + c0c), cB = ld4(a.cB + c0c), cC = ld4(a.cC + c0c);
    const float4 sc = ld4(a.sc + c0c), sh = ld4(a.sh + c0c), mu = ld4(a.mu + c0c), rs = ld4(a.rs + c0c);
    const float lo = a.act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = a.act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();

    // branch-free row loads: clamped coordinates, the select happens when the value is used
    auto load_row = [&](int t, float4& g, float4& v, float4& hg, float4& hv) {
        const int tc = t < 0 ? 0 : (t < Hs ? t : Hs - 1);
        g = ld4(a.dy + pix(tc, jc));
        v = ld4(a.zd + pix(tc, jc));
        if (halo) { hg = ld4(a.dy + pix(tc, hjc)); hv = ld4(a.zd + pix(tc, hjc)); }
    };
    auto dz_of = [&](const float4& g, const float4& v) { return add4_pk(add4_pk(mul4_pk(cA, g), cB), mul4_pk(cC, v)); };   // as bn_bwd_apply_kernel

    float4 g_cur, v_cur, hg_cur = zero4, hv_cur = zero4, ze_cur = zero4;
    load_row(i0 - 1, g_cur, v_cur, hg_cur, hv_cur);
    if (i1 > i0) {
        for (int t = i0 - 1; t <= i1; ++t) {
            // requests of the next iteration: the row to stage then, and z_e of the row formed then (output row t)
            float4 g_nxt, v_nxt, hg_nxt = zero4, hv_nxt = zero4, ze_nxt;
            load_row(t + 1, g_nxt, v_nxt, hg_nxt, hv_nxt);
            {
                const int oc = t < i0 ? i0 : (t < i1 ? t : i1 - 1);
                ze_nxt = ld4(a.ze + pix(oc, jc));
            }
            // ---- stage row t of dz_d (zeros outside the sub-image: the transposed conv's border)
            const bool row_ok = t >= 0 && t < Hs;
            const int slot = (t + 4) & 3;
            {
                const float4 d = dz_of(g_cur, v_cur);
                if (tcol < a.tw) ring[(slot * LC + tcol + 1) * kDg2Cg + tcg] = (row_ok && col_ok) ? d : zero4;
                if (halo) {
                    const float4 hd = dz_of(hg_cur, hv_cur);
                    ring[(slot * LC + hlc) * kDg2Cg + tcg] = (row_ok && hcol_ok) ? hd : zero4;
                }
            }
            __syncthreads();
            // ---- output row o = t - 1 from ring rows o - 1, o, o + 1
            const int o = t - 1;
            if (o >= i0 && o < i1 && col_ok) {
                float4 acc = zero4;
                const float4 y = muladd4_pk(ze_cur, sc, sh);
                const float4 ae = make_float4(__builtin_amdgcn_fmed3f(y.x, lo, hi), __builtin_amdgcn_fmed3f(y.y, lo, hi),
                                              __builtin_amdgcn_fmed3f(y.z, lo, hi), __builtin_amdgcn_fmed3f(y.w, lo, hi));
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int sl = (o - 1 + i + 4) & 3;
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const float4 v = ring[(sl * LC + tcol + jj) * kDg2Cg + tcg];
                        const float4 w4 = wv[i * 3 + jj];
                        acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                        acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
                        fma4_pk(dwv[8 - (i * 3 + jj)], ae, v);          // position (i, j) of the flipped kernel is tap 8 - (3i + j) of the forward conv
                    }
                }
                const float4 d = make_float4((y.x > lo && y.x < hi) ? acc.x : 0.f, (y.y > lo && y.y < hi) ? acc.y : 0.f,
                                             (y.z > lo && y.z < hi) ? acc.z : 0.f, (y.w > lo && y.w < hi) ? acc.w : 0.f);
                st4(a.out + pix(o, j), d);
                s1 = add4_pk(s1, d);
                s2 = add4_pk(s2, mul4_pk(mul4_pk(d, sub4_pk(ze_cur, mu)), rs));
            }
            g_cur = g_nxt; v_cur = v_nxt; hg_cur = hg_nxt; hv_cur = hv_nxt; ze_cur = ze_nxt;
        }
    }
    __syncthreads();                                                   // the ring is dead: its memory takes the block reduction
    float* sa = smem + tid * (NQ + 1);
    sa[0] = s1.x; sa[1] = s1.y; sa[2] = s1.z; sa[3] = s1.w; sa[4] = s2.x; sa[5] = s2.y; sa[6] = s2.z; sa[7] = s2.w;
#pragma unroll
    for (int k = 0; k < 9; ++k) { sa[8 + 4 * k] = dwv[k].x; sa[9 + 4 * k] = dwv[k].y; sa[10 + 4 * k] = dwv[k].z; sa[11 + 4 * k] = dwv[k].w; }
    __syncthreads();
    // the block's slice of its partial row: [sum dy | sum dy xhat | dW taps 0..8] x the 64 channels of this chunk; the 16 columns of a
    // channel group are added in ascending order (threads that never formed an output hold zeros)
    const int64_t row = (((int64_t)b * (R * R) + sub) * a.bands + band) * a.tiles_x + tx;
    float* prow = a.part + row * 11 * a.C;
    for (int e = tid; e < 11 * 4 * kDg2Cg; e += 256) {
        const int qn = e / (4 * kDg2Cg), c = e - qn * (4 * kDg2Cg);
        const int cg = c >> 2, comp = c & 3;
        const int ch = cc * (4 * kDg2Cg) + c;
        if (ch >= a.C) continue;
        float s = 0.f;
#pragma unroll
        for (int col = 0; col < kDg2Cols; ++col) s += smem[(col * kDg2Cg + cg) * (NQ + 1) + qn * 4 + comp];
        prow[(int64_t)qn * a.C + ch] = s;
    }
}

// geometry: column strips of <= 16 sub-image columns of (nearly) equal width; row bands until the launch has ~1024 blocks
static void dg2_plan(int B, int H, int W, int C, int rate, Dg2Args* a) {
    const int Hs = (H + rate - 1) / rate, Ws = (W + rate - 1) / rate;          // the largest parity class
    a->tiles_x = cdiv(Ws, kDg2Cols);
    a->tw = cdiv(Ws, a->tiles_x);
    a->cchunks = cdiv(C, 4 * kDg2Cg);
    const int64_t base = (int64_t)B * rate * rate * a->tiles_x * a->cchunks;
    int bands = (int)cdiv64(1024, base);
    const int max_bands = Hs / 4 > 0 ? Hs / 4 : 1;                             // at least four rows per band (each band re-stages two halo rows)
    bands = bands < 1 ? 1 : (bands > max_bands ? max_bands : bands);
    a->rows_per_band = cdiv(Hs, bands);
    a->bands = cdiv(Hs, a->rows_per_band);
}

size_t depthwise_dgrad_bn2_scratch(int B, int H, int W, int C, int rate) {
    if (C % 4 != 0 || (rate != 1 && rate != 2)) return (size_t)-1;
    Dg2Args a;
    dg2_plan(B, H, W, C, rate, &a);
    return (size_t)B * rate * rate * a.bands * a.tiles_x * 11 * (size_t)C;
}

// out [B,H,W,C] = dwconv3x3^T(cA dy + cB + cC zd, w) . act'(ze sc + sh) (stride 1, rate 1 | 2); partial rows [rows][11][C] of
// (sum out, sum out xhat_e, the nine taps sum act(ze sc + sh) . dz_d) in scratch, *rows_out = rows
int launch_depthwise_dgrad_bn2(const float* dy, const float* zd, const float* cA, const float* cB, const float* cC, int B, int H, int W, int C,
                               const float* w, int rate, const float* ze, const float* scale, const float* shift, int act, const float* mean,
                               const float* rstd, float* out, float* scratch, int* rows_out, hipStream_t st) {
    AMS_REQUIRE(C % 4 == 0 && C >= 4 && (rate == 1 || rate == 2) && B > 0 && H > 0 && W > 0, "depthwise_dgrad_bn2: bad shape C=%d rate=%d", C, rate);
    AMS_REQUIRE((int64_t)B * H * W * C < (int64_t)1 << 40, "depthwise_dgrad_bn2: tensor too large");
    Dg2Args a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.zd = zd; a.cA = cA; a.cB = cB; a.cC = cC; a.w = w; a.ze = ze; a.sc = scale; a.sh = shift; a.mu = mean; a.rs = rstd; a.act = act;
    a.out = out; a.part = scratch; a.B = B; a.H = H; a.W = W; a.C = C; a.R = rate;
    dg2_plan(B, H, W, C, rate, &a);
    const int64_t rows = (int64_t)B * rate * rate * a.bands * a.tiles_x;
    const int64_t nb = rows * a.cchunks;
    AMS_REQUIRE(nb > 0 && nb < 0x7fffffffLL, "depthwise_dgrad_bn2: bad grid");
    *rows_out = (int)rows;
    note_kernel(rate == 2 ? "dw3x3_dgrad_bn2_kernel<2>" : "dw3x3_dgrad_bn2_kernel<1>");
    if (rate == 1) hipLaunchKernelGGL((dw3x3_dgrad_bn2_kernel<1>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw3x3_dgrad_bn2_kernel<2>), dim3((unsigned)nb), dim3(256), 0, st, a);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

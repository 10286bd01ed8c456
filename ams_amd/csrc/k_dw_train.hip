// Fine-tune step, stride-16 blocks: the depthwise layer's BN-backward apply pass (dz_d = A dy + B + C z_d, three tensor passes of up to
// 66 MB each at 8 x 512x1024) folded into the one-kernel depthwise backward of k_conv.hip (dw3x3_dgrad_bn_kernel).
//
// That kernel reads its nine taps from global memory, three columns per thread: applying (A, B, C) on those loads costs two loads and
// five VALU per tap, three times redundant along a row — measured no faster than the separate pass.  Here the
// taps go through LDS instead, so every dz_d value is formed ONCE per tile:
//   * a block owns one image, one parity sub-image (rate 2 = four independent rate-1 problems on the (row, column) parity classes), a
//     tile of <= 8 rows x <= 16 columns of it and a 64-channel chunk (256 contiguous bytes per pixel);
//   * all loads of the tile are requested up front — (dy, z_d) of the tile + a one-pixel halo, z_e of the thread's own outputs — then
//     dz_d = (A dy + B) + C z_d (the unfused operations of bn_bwd_apply_kernel: bit-identical to the pass it replaces; zeros outside the
//     sub-image) goes into an LDS tile, ONE barrier, and thread (column, 4 channels) walks down its <= 8 outputs: da_e = dwconv^T(dz_d)
//     with the same tap order and fmaf chain as dw3x3_dgrad_bn_kernel (bit-identical result), x act'(z_e sc + sh), the BN-backward sums
//     of the expand layer and the nine taps of the depthwise weight gradient in registers as there.
// (A first form that streamed rows through a four-row ring with one barrier per row was latency-bound: 170-213 us at 960 channels against
// 132 us for the two kernels it replaced.)
// One partial row [11][C] per (image, sub-image, tile), each block writing the 64-channel slice it owns; threads that share a channel
// group are folded in a fixed order through LDS (deterministic).
#include "kernels.hpp"

namespace ams {

struct Dg2Args {
    const float* dy; const float* zd; const float* cA; const float* cB; const float* cC;      // depthwise layer: masked gradient, raw output, BN-backward coefficients
    const float* w;                                                                           // depthwise weights [9][C]
    const float* ze; const float* sc; const float* sh; const float* mu; const float* rs; int act;     // expand layer: raw output, BN scale / shift / mean / rstd
    float* out; float* part;
    int B, H, W, C, R;
    int tiles_x, tw, tiles_y, th, cchunks;
};

constexpr int kDg2Cols = 16, kDg2Rows = 9, kDg2Cg = 16;          // most columns and rows of a block's tile; channel groups (x 4 channels) of its chunk
constexpr int kDg2Pix = 180;                                     // pixels of the haloed tile that fit the LDS buffer: (th + 2)(tw + 2) <= 180

template <int R>
__global__ __launch_bounds__(256, 3) void dw3x3_dgrad_bn2_kernel(Dg2Args a) {
    constexpr int NQ = 8 + 36;                                      // s1, s2, nine taps: 4 channels each
    constexpr int NST = (kDg2Pix * kDg2Cg + 255) / 256;             // staging duties per thread (12)
    static_assert(NST % 2 == 0, "staged in two rounds");
    // the dz_d tile (<= 180 pixels x 16 float4 = 46 080 B) and, after the walk, the block reduction's 256 x 45 floats (the same 46 080 B)
    __shared__ __attribute__((aligned(16))) float smem[256 * (NQ + 1)];
    static_assert(kDg2Pix * kDg2Cg * 4 <= 256 * (NQ + 1), "the tile must fit the reduction buffer");
    float4* tile = reinterpret_cast<float4*>(smem);

    int bid = blockIdx.x;
    const int cc = bid % a.cchunks; bid /= a.cchunks;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int sub = bid % (R * R);
    const int b = bid / (R * R);
    const int py = sub / R, px = sub - py * R;
    const int Hs = (a.H - py + R - 1) / R, Ws = (a.W - px + R - 1) / R;
    const int tid = threadIdx.x, tcol = tid >> 4, tcg = tid & 15;
    const int cbase = cc * (4 * kDg2Cg);
    const int i0 = ty * a.th, j0 = tx * a.tw;
    const int64_t img = (int64_t)b * a.H * a.W * a.C;
    auto pix = [&](int i, int jj, int c) { return img + ((int64_t)(py + R * i) * a.W + (px + R * jj)) * a.C + c; };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // this thread's channel group: the same for its staging duties (e & 15 == tid & 15) and for its outputs
    const int c0 = cbase + 4 * tcg;
    const bool chan_ok = c0 < a.C;
    const int c0c = chan_ok ? c0 : a.C - 4;
    const int j = j0 + tcol;
    const bool col_ok = tcol < a.tw && j < Ws && chan_ok;
    const int jcl = j < Ws ? j : Ws - 1;
    const float4 cA = ld4(a.cA + c0c), cB = ld4(a.cB + c0c), cC = ld4(a.cC + c0c);
    const int lcw = a.tw + 2, lrh = a.th + 2;                      // haloed tile; lcw is also the LDS pitch
    // ---- the loads of the tile in two rounds of NST / 2 staging duties (dy and z_d each): duty e = tid + 256 u -> (row r, column c) of the
    // haloed tile; dz_d = (A g + B) + C z as bn_bwd_apply_kernel forms it; zeros outside the sub-image (the transposed conv's border)
    float4 zev[kDg2Rows];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 sg[NST / 2], sv[NST / 2];
#pragma unroll
        for (int u = 0; u < NST / 2; ++u) {
            const int rc = (tid + 256 * (half * (NST / 2) + u)) >> 4;
            const int r = rc / lcw, c = rc - r * lcw;
            const int i = i0 - 1 + r, jj = j0 - 1 + c;
            const int ic = i < 0 ? 0 : (i < Hs ? i : Hs - 1), jc = jj < 0 ? 0 : (jj < Ws ? jj : Ws - 1);
            sg[u] = ld4(a.dy + pix(ic, jc, c0c));
            sv[u] = ld4(a.zd + pix(ic, jc, c0c));
        }
        if (half == 0) {                                           // z_e of the thread's own outputs rides behind the first round
#pragma unroll
            for (int r = 0; r < kDg2Rows; ++r) {
                const int i = i0 + r;
                zev[r] = ld4(a.ze + pix(i < Hs ? i : Hs - 1, jcl, c0c));
            }
        }
#pragma unroll
        for (int u = 0; u < NST / 2; ++u) {
            const int rc = (tid + 256 * (half * (NST / 2) + u)) >> 4;
            const int r = rc / lcw, c = rc - r * lcw;
            const int i = i0 - 1 + r, jj = j0 - 1 + c;
            const bool ok = i >= 0 && i < Hs && jj >= 0 && jj < Ws && chan_ok;
            const float4 d = add4_pk(add4_pk(mul4_pk(cA, sg[u]), cB), mul4_pk(cC, sv[u]));
            if (r < lrh) tile[(r * lcw + c) * kDg2Cg + tcg] = make_float4(ok ? d.x : 0.f, ok ? d.y : 0.f, ok ? d.z : 0.f, ok ? d.w : 0.f);      // (a float4 ?: goes through scratch memory)
        }
    }
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(a.w + (8 - k) * a.C + c0c);          // the conv with the flipped kernel
    const float4 sc = ld4(a.sc + c0c), sh = ld4(a.sh + c0c), mu = ld4(a.mu + c0c), rs = ld4(a.rs + c0c);
    const float lo = a.act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = a.act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    __syncthreads();

    // ---- thread (column, 4 channels) walks down the tile's rows
    float4 s1 = zero4, s2 = zero4, dwv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv[k] = zero4;
    if (col_ok) {
#pragma unroll
        for (int r = 0; r < kDg2Rows; ++r) {
            const int o = i0 + r;
            if (r >= a.th || o >= Hs) break;
            float4 acc = zero4;
            const float4 y = muladd4_pk(zev[r], sc, sh);
            const float4 ae = make_float4(__builtin_amdgcn_fmed3f(y.x, lo, hi), __builtin_amdgcn_fmed3f(y.y, lo, hi),
                                          __builtin_amdgcn_fmed3f(y.z, lo, hi), __builtin_amdgcn_fmed3f(y.w, lo, hi));
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const float4 v = tile[((r + i) * lcw + tcol + jj) * kDg2Cg + tcg];
                    const float4 w4 = wv[i * 3 + jj];
                    acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                    acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
                    fma4_pk(dwv[8 - (i * 3 + jj)], ae, v);              // position (i, j) of the flipped kernel is tap 8 - (3i + j) of the forward conv
                }
            const float4 d = make_float4((y.x > lo && y.x < hi) ? acc.x : 0.f, (y.y > lo && y.y < hi) ? acc.y : 0.f,
                                         (y.z > lo && y.z < hi) ? acc.z : 0.f, (y.w > lo && y.w < hi) ? acc.w : 0.f);
            st4(a.out + pix(o, j, c0), d);
            s1 = add4_pk(s1, d);
            s2 = add4_pk(s2, mul4_pk(mul4_pk(d, sub4_pk(zev[r], mu)), rs));
        }
    }
    __syncthreads();                                                   // the tile is dead: its memory takes the block reduction
    float* sa = smem + tid * (NQ + 1);
    sa[0] = s1.x; sa[1] = s1.y; sa[2] = s1.z; sa[3] = s1.w; sa[4] = s2.x; sa[5] = s2.y; sa[6] = s2.z; sa[7] = s2.w;
#pragma unroll
    for (int k = 0; k < 9; ++k) { sa[8 + 4 * k] = dwv[k].x; sa[9 + 4 * k] = dwv[k].y; sa[10 + 4 * k] = dwv[k].z; sa[11 + 4 * k] = dwv[k].w; }
    __syncthreads();
    // the block's slice of its partial row: [sum dy | sum dy xhat | dW taps 0..8] x the 64 channels of this chunk; the 16 columns of a
    // channel group are added in ascending order (threads that never formed an output hold zeros)
    const int64_t row = (((int64_t)b * (R * R) + sub) * a.tiles_y + ty) * a.tiles_x + tx;
    float* prow = a.part + row * 11 * a.C;
    for (int e = tid; e < 11 * 4 * kDg2Cg; e += 256) {
        const int qn = e / (4 * kDg2Cg), c = e - qn * (4 * kDg2Cg);
        const int cg = c >> 2, comp = c & 3;
        const int ch = cbase + c;
        if (ch >= a.C) continue;
        float s = 0.f;
#pragma unroll
        for (int col = 0; col < kDg2Cols; ++col) s += smem[(col * kDg2Cg + cg) * (NQ + 1) + qn * 4 + comp];
        prow[(int64_t)qn * a.C + ch] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Training FORWARD of the same layers in the same tile form (replaces dw3x3_fwd_bn_kernel, AMS_OPT_FUSE_DGRAD_BN = 3): a_e = act(z_e sc + sh)
// is formed once per element of the haloed tile on its way into LDS (the old kernel applies BN + activation to each of its 4.5 tap loads
// per output), one barrier, z_d = dwconv(a_e) with the tap order and fmaf chain of that kernel (bit-identical z_d), and the shifted sums of
// z_d for its BN statistics.
// PERSISTENT blocks (round 4): a block keeps one 64-channel chunk and walks a fixed sequence of tiles.  One block per tile measured 59 us at
// 960 channels of which 28 us remained with every tensor load, LDS write, tap and store removed: three dependent round trips of block
// set-up (arguments, the chunk's vectors and weights, the partial row), three barriers and a block reduction PER TILE on 2880 blocks.  Now
// the chunk's constants are loaded once, the sums stay in registers across tiles (one reduction, one partial row [2][C] per block), and the
// next tile's loads are in flight while the current one is walked.
struct Df2Args {
    const float* ze; const float* sc; const float* sh; int act;
    const float* w; const float* center;
    float* zd; float* part;
    int B, H, W, C, R;
    int tiles_x, tw, tiles_y, th, cchunks;
    int n_tiles;                       // B R^2 tiles_y tiles_x; block (p, chunk) walks tiles p, p + P, ...  (P = gridDim.x / cchunks)
};

struct Dg2Tile { int b, py, px, Hs, Ws, i0, j0; };
template <int R>
__device__ __forceinline__ Dg2Tile dg2_tile(int t, int tiles_x, int tiles_y, int tw, int th, int H, int W) {      // block-uniform
    Dg2Tile g;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int sub = t % (R * R);
    g.b = t / (R * R);
    g.py = sub / R; g.px = sub - g.py * R;
    g.Hs = (H - g.py + R - 1) / R; g.Ws = (W - g.px + R - 1) / R;
    g.i0 = ty * th; g.j0 = tx * tw;
    return g;
}

template <int R>
__global__ __launch_bounds__(256, 2) void dw3x3_fwd_bn2_kernel(Df2Args a) {
    constexpr int NST = (kDg2Pix * kDg2Cg + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];       // the haloed tile ((th + 2)(tw + 2) x 16 float4), then the block reduction (256 x 9 floats)
    float4* tile = reinterpret_cast<float4*>(smem);
    const int cc = blockIdx.x % a.cchunks, p = blockIdx.x / a.cchunks, P = gridDim.x / a.cchunks;
    const int tid = threadIdx.x, tcol = tid >> 4, tcg = tid & 15;
    const int cbase = cc * (4 * kDg2Cg);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c0 = cbase + 4 * tcg;
    const bool chan_ok = c0 < a.C;
    const int c0c = chan_ok ? c0 : a.C - 4;
    const float4 sc = ld4(a.sc + c0c), sh = ld4(a.sh + c0c);
    const float lo = a.act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = a.act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    const int lcw = a.tw + 2, lrh = a.th + 2;
    // staging duty u of this thread = haloed-tile position (tid >> 4) + 16 u, row-major with pitch lcw: the same for every tile
    const int q16 = 16 / lcw, r16 = 16 - q16 * lcw;
    const int r_first = tcol / lcw, c_first = tcol - r_first * lcw;
    const int rowC = R * a.W * a.C, colC = R * a.C;                    // element steps of one sub-image row / column (32-bit: checked on the host)
    float4 sv[NST];
    auto issue = [&](const Dg2Tile& g) {
        const float* base = a.ze + (int64_t)g.b * a.H * a.W * a.C + (g.py * a.W + g.px) * a.C + c0c;
        int r = r_first, c = c_first;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = g.i0 - 1 + r, jj = g.j0 - 1 + c;
            const int ic = i < 0 ? 0 : (i < g.Hs ? i : g.Hs - 1), jc = jj < 0 ? 0 : (jj < g.Ws ? jj : g.Ws - 1);
            sv[u] = ld4(base + ic * rowC + jc * colC);
            c += r16; r += q16;
            if (c >= lcw) { c -= lcw; ++r; }
        }
    };
    int t = p;
    if (t < a.n_tiles) issue(dg2_tile<R>(t, a.tiles_x, a.tiles_y, a.tw, a.th, a.H, a.W));
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4(a.w + k * a.C + c0c);
    const float4 ctr = a.center ? ld4(a.center + c0c) : zero4;
    float4 s1 = zero4, s2 = zero4;
    for (; t < a.n_tiles; t += P) {
        const Dg2Tile g = dg2_tile<R>(t, a.tiles_x, a.tiles_y, a.tw, a.th, a.H, a.W);
        {
            int r = r_first, c = c_first;
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int i = g.i0 - 1 + r, jj = g.j0 - 1 + c;
                const bool ok = i >= 0 && i < g.Hs && jj >= 0 && jj < g.Ws && chan_ok;      // SAME padding pads the ACTIVATION with zeros
                const float4 y = muladd4_pk(sv[u], sc, sh);
                const float4 v = make_float4(__builtin_amdgcn_fmed3f(y.x, lo, hi), __builtin_amdgcn_fmed3f(y.y, lo, hi), __builtin_amdgcn_fmed3f(y.z, lo, hi),
                                             __builtin_amdgcn_fmed3f(y.w, lo, hi));
                if (r < lrh) tile[(r * lcw + c) * kDg2Cg + tcg] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);      // (a float4 ?: goes through scratch memory)
                c += r16; r += q16;
                if (c >= lcw) { c -= lcw; ++r; }
            }
        }
        __syncthreads();
        if (t + P < a.n_tiles) issue(dg2_tile<R>(t + P, a.tiles_x, a.tiles_y, a.tw, a.th, a.H, a.W));      // in flight across the walk
        const int j = g.j0 + tcol;
        const int n_rows = g.Hs - g.i0 < a.th ? g.Hs - g.i0 : a.th;                                           // block-uniform
        if (tcol < a.tw && j < g.Ws && chan_ok) {
            float* outp = a.zd + (int64_t)g.b * a.H * a.W * a.C + ((g.py + R * g.i0) * a.W + (g.px + R * j)) * a.C + c0;
#pragma unroll
            for (int r = 0; r < kDg2Rows; ++r) {
                if (r >= n_rows) break;
                float4 acc = zero4;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const float4 v = tile[((r + i) * lcw + tcol + jj) * kDg2Cg + tcg];
                        const float4 w4 = wv[i * 3 + jj];
                        acc.x = fmaf(v.x, w4.x, acc.x); acc.y = fmaf(v.y, w4.y, acc.y);
                        acc.z = fmaf(v.z, w4.z, acc.z); acc.w = fmaf(v.w, w4.w, acc.w);
                    }
                st4(outp + r * rowC, acc);
                const float4 d = sub4_pk(acc, ctr);
                s1 = add4_pk(s1, d);
                s2 = add4_pk(s2, mul4_pk(d, d));
            }
        }
        __syncthreads();                                               // the tile is consumed: the next one (or the reduction) may overwrite it
    }
    float* sa = smem + tid * 9;
    sa[0] = s1.x; sa[1] = s1.y; sa[2] = s1.z; sa[3] = s1.w; sa[4] = s2.x; sa[5] = s2.y; sa[6] = s2.z; sa[7] = s2.w;
    __syncthreads();
    // the block's 64-channel slice of ITS partial row: the 16 columns of a channel group are added in ascending order (deterministic)
    float* prow = a.part + (int64_t)p * 2 * a.C;
    for (int e = tid; e < 2 * 4 * kDg2Cg; e += 256) {
        const int qn = e / (4 * kDg2Cg), c = e - qn * (4 * kDg2Cg);
        const int cg = c >> 2, comp = c & 3;
        const int ch = cbase + c;
        if (ch >= a.C) continue;
        float s = 0.f;
#pragma unroll
        for (int col = 0; col < kDg2Cols; ++col) s += smem[(col * kDg2Cg + cg) * 9 + qn * 4 + comp];
        prow[(int64_t)qn * a.C + ch] = s;
    }
}

// geometry: tiles of <= 9 x <= 16 sub-image pixels of (nearly) equal size whose haloed form fits the LDS buffer
static void dg2_plan(int B, int H, int W, int C, int rate, Dg2Args* a) {
    const int Hs = (H + rate - 1) / rate, Ws = (W + rate - 1) / rate;          // the largest parity class
    a->tiles_y = cdiv(Hs, kDg2Rows);
    a->th = cdiv(Hs, a->tiles_y);
    int tw_max = kDg2Pix / (a->th + 2) - 2;                                    // (th + 2)(tw + 2) pixels of dz_d in LDS
    if (tw_max > kDg2Cols) tw_max = kDg2Cols;
    a->tiles_x = cdiv(Ws, tw_max);
    a->tw = cdiv(Ws, a->tiles_x);
    a->cchunks = cdiv(C, 4 * kDg2Cg);
}

size_t depthwise_dgrad_bn2_scratch(int B, int H, int W, int C, int rate) {
    if (C % 4 != 0 || C < 4 || (rate != 1 && rate != 2)) return (size_t)-1;
    Dg2Args a;
    dg2_plan(B, H, W, C, rate, &a);
    return (size_t)B * rate * rate * a.tiles_y * a.tiles_x * 11 * (size_t)C;
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
    const int64_t rows = (int64_t)B * rate * rate * a.tiles_y * a.tiles_x;
    const int64_t nb = rows * a.cchunks;
    AMS_REQUIRE(nb > 0 && nb < 0x7fffffffLL, "depthwise_dgrad_bn2: bad grid");
    *rows_out = (int)rows;
    note_kernel(rate == 2 ? "dw3x3_dgrad_bn2_kernel<2>" : "dw3x3_dgrad_bn2_kernel<1>");
    if (rate == 1) hipLaunchKernelGGL((dw3x3_dgrad_bn2_kernel<1>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw3x3_dgrad_bn2_kernel<2>), dim3((unsigned)nb), dim3(256), 0, st, a);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

size_t depthwise_fwd_bn2_scratch(int B, int H, int W, int C, int rate) {
    if (C % 4 != 0 || C < 4 || (rate != 1 && rate != 2)) return (size_t)-1;
    if ((int64_t)H * W * C >= 0x7fffffffLL) return (size_t)-1;                 // 32-bit element offsets inside an image: such a map takes the older kernel
    Dg2Args a;
    dg2_plan(B, H, W, C, rate, &a);
    return (size_t)B * rate * rate * a.tiles_y * a.tiles_x * 2 * (size_t)C;
}

// blocks per channel chunk of a persistent launch: as many as are co-resident, then fewer so that every block walks the same number of tiles
static int dg2_blocks_per_chunk(int64_t n_tiles, int cchunks, int64_t slots) {
    int64_t per = slots / cchunks;
    if (per < 1) per = 1;
    if (per > n_tiles) per = n_tiles;
    const int64_t k = cdiv64(n_tiles, per);              // tiles per block
    return (int)cdiv64(n_tiles, k);
}

// zd [B,H,W,C] = dwconv3x3(act(ze scale + shift), w) (stride 1, rate 1 | 2), partial rows [rows][2][C] of (sum (zd - center), sum (zd - center)^2)
int launch_depthwise_fwd_bn2(const float* ze, int B, int H, int W, int C, const float* w, int rate, const float* scale, const float* shift, int act,
                             const float* center, float* zd, float* scratch, int* rows_out, hipStream_t st) {
    AMS_REQUIRE(C % 4 == 0 && C >= 4 && (rate == 1 || rate == 2) && B > 0 && H > 0 && W > 0, "depthwise_fwd_bn2: bad shape C=%d rate=%d", C, rate);
    AMS_REQUIRE((int64_t)H * W * C < 0x7fffffffLL, "depthwise_fwd_bn2: image too large for 32-bit element offsets");
    Dg2Args g;
    dg2_plan(B, H, W, C, rate, &g);
    Df2Args a;
    memset(&a, 0, sizeof(a));
    a.ze = ze; a.sc = scale; a.sh = shift; a.act = act; a.w = w; a.center = center; a.zd = zd; a.part = scratch;
    a.B = B; a.H = H; a.W = W; a.C = C; a.R = rate;
    a.tiles_x = g.tiles_x; a.tw = g.tw; a.tiles_y = g.tiles_y; a.th = g.th; a.cchunks = g.cchunks;
    const int64_t n_tiles = (int64_t)B * rate * rate * a.tiles_y * a.tiles_x;
    AMS_REQUIRE(n_tiles > 0 && n_tiles * a.cchunks < 0x7fffffffLL, "depthwise_fwd_bn2: bad grid");
    a.n_tiles = (int)n_tiles;
    size_t lds = (size_t)(a.th + 2) * (a.tw + 2) * kDg2Cg * 16;
    if (lds < 256 * 9 * sizeof(float)) lds = 256 * 9 * sizeof(float);
    const void* fn = rate == 1 ? (const void*)dw3x3_fwd_bn2_kernel<1> : (const void*)dw3x3_fwd_bn2_kernel<2>;
    int per_cu = 1, cus = 256;
    RUN_RC(func_blocks_per_cu(fn, 256, lds, &per_cu));
    RUN_RC(device_cus(&cus));
    const int P = dg2_blocks_per_chunk(n_tiles, a.cchunks, (int64_t)per_cu * cus);
    *rows_out = P;
    note_kernel(rate == 2 ? "dw3x3_fwd_bn2_kernel<2>" : "dw3x3_fwd_bn2_kernel<1>");
    if (rate == 1) hipLaunchKernelGGL((dw3x3_fwd_bn2_kernel<1>), dim3((unsigned)(P * a.cchunks)), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((dw3x3_fwd_bn2_kernel<2>), dim3((unsigned)(P * a.cchunks)), dim3(256), lds, st, a);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

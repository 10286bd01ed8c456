// Output head of the AMS student: the full-resolution logits (B x H x W x 19 f32 = 40 MB per 512x1024 frame)
// are never materialised.  One kernel interpolates the output-stride-16 logits (ResizeBilinear, align_corners),
// gathers the K selected classes, takes the argmax and, when teacher labels are given, accumulates the K x K
// confusion matrix and the masked cross-entropy sums (reference utils/graph_utils.py:373-408,
// SemanticNetwork.py:96-115).  The backward kernel is the exact transpose: every low-resolution logit gathers
// (softmax - onehot)/N from the full-resolution pixels it was interpolated into, in a fixed order.
#include "kernels.hpp"

namespace ams {

constexpr int kMaxK = 32;

struct HeadGeom {
    int B, h, w, ld, K, H, W, NC;
    int per_frame;         // metrics per frame: conf [B][K][K], loss [B][2] instead of the batch totals
    int labels_u8;         // the label map as uint8 [B][H][W] through the same pointer (K <= 32 fits a byte: a quarter of the device -> host bytes)
    float sy, sx;          // (h-1)/(H-1), (w-1)/(W-1) as f32 (TF: CalculateResizeScale with align_corners)
};

__device__ __forceinline__ void src_tap(int dst, float scale, int n_in, int& lo, int& hi, float& t) {
    const float src = __fmul_rn((float)dst, scale);
    const float fl = floorf(src);
    lo = (int)fl;
    hi = lo + 1 < n_in ? lo + 1 : n_in - 1;
    t = __fsub_rn(src, fl);
}

// v = top + (bot - top) * ty,  top = tl + (tr - tl) * tx   (unfused, like the TF CPU kernel / the oracle)
__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, float tx, float ty) {
    const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), tx));
    const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), tx));
    return __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ty));
}

// Soft-teacher targets (create_student_v3 with soft_teacher=True, utils/graph_utils.py:359, 375-376, 403-404): teacher logits
// [B][th][tw][ld] f32 fed through teacher_labels_logits_pl; the target of a pixel is softmax(gather(teacher_logits, class_weights)).  th x tw is
// either the label size (the reference's feed: the loss needs the shape of filtered_logits) or any smaller grid, which is then interpolated
// to H x W exactly as the student's own logits are (align corners) — at th == H, tw == W that interpolation is the identity, bit for bit.
struct SoftTeacher {
    const float* t;
    int th, tw, ld;
    float sy, sx;
};

struct ClassTable {
    int32_t idx[kMaxK];      // selected class ids
    int32_t lut[256];        // teacher id -> subset index, -1 = ignored
};

// KMAX > 0: K <= KMAX and the K horizontally interpolated values of the two source rows live in registers while the thread walks
// DOWN its column through a band of consecutive output rows — they change only when the source row does (every ~16 output rows at
// 512 from 33), so an output pixel costs K vertical lerps + the argmax instead of 4K loads and 3K lerps.  Same arithmetic in the same
// order per pixel (horizontal, then vertical, unfused): bit-identical labels.  KMAX = 0: any K <= kMaxK, everything per pixel.
template <int KMAX>
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* __restrict__ logits, HeadGeom g, ClassTable ct,
                                                              const uint8_t* __restrict__ teacher,
                                                              int32_t* __restrict__ labels,
                                                              unsigned long long* __restrict__ conf, double* __restrict__ loss) {
    __shared__ int s_conf[kMaxK * kMaxK];
    __shared__ double s_loss[4];
    __shared__ int s_cnt[4];
    const bool metric = teacher != nullptr;
    if (metric)
        for (int e = threadIdx.x; e < g.K * g.K; e += blockDim.x) s_conf[e] = 0;
    const int b = blockIdx.z;
    if (g.per_frame && metric) { conf += (size_t)b * g.K * g.K; loss += 2 * b; }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (metric) __syncthreads();
    // A pixel's loss enters the sums as an integer multiple of 2^-20 (held in f64: exact up to 2^33): every partial sum is then exact,
    // so the total does not depend on how pixels are dealt to threads, blocks or batches, nor on the order of the atomics below — the
    // loss of a frame is the same bits in a one-frame call and inside a 32-frame call.  The rounding is <= 5e-7 per pixel (~1e-10 of the sum).
    double my_loss = 0.0;
    int my_cnt = 0;
    int x0 = 0, x1 = 0; float tx = 0.f;
    if (x < g.W) src_tap(x, g.sx, g.w, x0, x1, tx);
    const float* base = logits + (int64_t)b * g.h * g.w * g.ld;
    // a block walks a band of consecutive rows: the confusion counts and the loss leave the block once (the global atomics on a
    // handful of addresses were the cost of the metric path with one row per block)
    const int band = (g.H + (int)gridDim.y - 1) / (int)gridDim.y;
    const int ybeg = blockIdx.y * band, yend = ybeg + band < g.H ? ybeg + band : g.H;
    constexpr int KR = KMAX > 0 ? KMAX : 1;
    float top[KR], bot[KR];
    int cur_y0 = -1;
    for (int y = ybeg; y < yend; ++y) {
        if (x >= g.W) break;
        int y0, y1; float ty;
        src_tap(y, g.sy, g.h, y0, y1, ty);
        const float* ptl = base + ((int64_t)y0 * g.w + x0) * g.ld;
        const float* ptr = base + ((int64_t)y0 * g.w + x1) * g.ld;
        const float* pbl = base + ((int64_t)y1 * g.w + x0) * g.ld;
        const float* pbr = base + ((int64_t)y1 * g.w + x1) * g.ld;
        if (KMAX > 0 && y0 != cur_y0) {               // block-uniform (one output row per iteration)
            cur_y0 = y0;
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int c = ct.idx[k < g.K ? k : 0];
                top[k] = __fadd_rn(ptl[c], __fmul_rn(__fsub_rn(ptr[c], ptl[c]), tx));
                bot[k] = __fadd_rn(pbl[c], __fmul_rn(__fsub_rn(pbr[c], pbl[c]), tx));
            }
        }
        const int64_t pix = ((int64_t)b * g.H + y) * g.W + x;
        int target = -1;
        if (metric) target = ct.lut[teacher[pix]];
        float best = 0.f, zt = 0.f, zmax = 0.f, ssum = 0.f;
        int arg = 0;
        auto visit = [&](int k, float v) {
            if (k == 0 || v > best) { best = v; arg = k; }        // first maximum wins (tf.argmax)
            if (target >= 0) {
                // streaming log-sum-exp: keep the running max, rescale the running sum
                if (k == 0) { zmax = v; ssum = 1.f; }
                else if (v > zmax) { ssum = ssum * __expf(zmax - v) + 1.f; zmax = v; }
                else ssum += __expf(v - zmax);
                if (k == target) zt = v;
            }
        };
        if (KMAX > 0) {
#pragma unroll
            for (int k = 0; k < KR; ++k)
                if (k < g.K) visit(k, __fadd_rn(top[k], __fmul_rn(__fsub_rn(bot[k], top[k]), ty)));
        } else {
            for (int k = 0; k < g.K; ++k) {
                const int c = ct.idx[k];
                visit(k, bilerp(ptl[c], ptr[c], pbl[c], pbr[c], tx, ty));
            }
        }
        if (labels) { if (g.labels_u8) reinterpret_cast<uint8_t*>(labels)[pix] = (uint8_t)arg; else labels[pix] = arg; }
        if (target >= 0) {
            my_loss += rint((double)((zmax + __logf(ssum)) - zt) * 1048576.0);
            my_cnt += 1;
            atomicAdd(&s_conf[target * g.K + arg], 1);
        }
    }
    if (!metric) return;
    my_loss = wave_sum(my_loss);
    my_cnt = (int)wave_sum((float)my_cnt);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_loss[wave] = my_loss; s_cnt[wave] = my_cnt; }
    __syncthreads();
    for (int e = threadIdx.x; e < g.K * g.K; e += blockDim.x)
        if (s_conf[e]) atomicAdd(&conf[e], (unsigned long long)s_conf[e]);
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        double ls = 0; int cn = 0;
        for (int i = 0; i < nw; ++i) { ls += s_loss[i]; cn += s_cnt[i]; }
        if (cn) { atomicAdd(&loss[0], ls * (1.0 / 1048576.0)); atomicAdd(&loss[1], (double)cn); }
    }
}

static int fill_class_table(const int32_t* cls_host, int K, int NC, ClassTable* ct) {
    AMS_REQUIRE(K > 0 && K <= kMaxK, "head: K=%d out of range (1..%d)", K, kMaxK);
    for (int i = 0; i < 256; ++i) ct->lut[i] = -1;
    for (int k = 0; k < kMaxK; ++k) ct->idx[k] = 0;
    for (int k = 0; k < K; ++k) {
        AMS_REQUIRE(cls_host[k] >= 0 && cls_host[k] < NC && cls_host[k] < 256, "head: class id %d out of range", cls_host[k]);
        ct->idx[k] = cls_host[k];
        ct->lut[cls_host[k]] = k;
    }
    return AMS_OK;
}

static HeadGeom head_geom(int ld, int B, int h, int w, int K, int H, int W, int NC) {
    HeadGeom g;
    g.B = B; g.h = h; g.w = w; g.ld = ld; g.K = K; g.H = H; g.W = W; g.NC = NC; g.per_frame = 0; g.labels_u8 = 0;
    g.sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    g.sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    return g;
}

// cls: HOST pointer to the K selected class ids (they travel to the kernel by value)
int launch_upsample_argmax(const float* logits, int ld, int B, int h, int w, const int32_t* cls, int K, int H, int W,
                           const uint8_t* teacher, int NC, int32_t* labels, int64_t* conf, double* loss, hipStream_t st, int per_frame, int labels_u8) {
    ClassTable ct;
    int rc = fill_class_table(cls, K, NC, &ct);
    if (rc) return rc;
    AMS_REQUIRE(teacher == nullptr || (conf != nullptr && loss != nullptr), "head: metrics need conf and loss buffers");
    const int nm = per_frame ? B : 1;
    if (teacher) {
        AMS_CHECK_HIP(hipMemsetAsync(conf, 0, sizeof(int64_t) * K * K * nm, st));
        AMS_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(double) * 2 * nm, st));
    }
    HeadGeom g = head_geom(ld, B, h, w, K, H, W, NC);
    g.per_frame = per_frame;
    g.labels_u8 = labels_u8;
    note_kernel("upsample_argmax_kernel");
    // bands of consecutive rows per block: 32 per column strip and image, fewer rows per band when that leaves the chip short of blocks
    int rows_y = H < 32 ? H : 32;
    while (rows_y < H && (int64_t)cdiv(W, 256) * rows_y * B < 2048) rows_y *= 2;
    if (rows_y > H) rows_y = H;
    const dim3 grid(cdiv(W, 256), rows_y, B);
    if (K <= 8)
        hipLaunchKernelGGL(upsample_argmax_kernel<8>, grid, dim3(256), 0, st, logits, g, ct, teacher, labels, (unsigned long long*)conf, loss);
    else
        hipLaunchKernelGGL(upsample_argmax_kernel<0>, grid, dim3(256), 0, st, logits, g, ct, teacher, labels, (unsigned long long*)conf, loss);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// d loss / d low-res logits.  One block per low-resolution cell (b, i, j): its threads walk the full-resolution
// pixels whose bilinear footprint contains the cell, recompute that pixel's K interpolated logits and softmax,
// and accumulate weight * (softmax_k - onehot_k) / N.  Fixed traversal + tree reduction = deterministic.
// ---------------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void ce_grad_kernel(const float* __restrict__ logits, HeadGeom g, ClassTable ct,
                                                      const uint8_t* __restrict__ teacher,
                                                      const double* __restrict__ loss_and_count, float* __restrict__ dlogits,
                                                      int ldd, float empty_val) {
    __shared__ float s_part[4][KMAX];
    const int j = blockIdx.x, i = blockIdx.y, b = blockIdx.z;
    const double nvalid = loss_and_count[1];
    const float inv_n = nvalid > 0.5 ? (float)(1.0 / nvalid) : 0.f;
    // candidate full-resolution rows / columns: everything whose source coordinate can fall in [i-1, i+1)
    const float ry = g.sy > 0.f ? 1.f / g.sy : 0.f, rx = g.sx > 0.f ? 1.f / g.sx : 0.f;
    int ya = g.sy > 0.f ? (int)floorf((i - 1) * ry) - 1 : 0, yb = g.sy > 0.f ? (int)ceilf((i + 1) * ry) + 1 : g.H - 1;
    int xa = g.sx > 0.f ? (int)floorf((j - 1) * rx) - 1 : 0, xb = g.sx > 0.f ? (int)ceilf((j + 1) * rx) + 1 : g.W - 1;
    ya = ya < 0 ? 0 : ya; xa = xa < 0 ? 0 : xa;
    yb = yb > g.H - 1 ? g.H - 1 : yb; xb = xb > g.W - 1 ? g.W - 1 : xb;
    const int ny = yb - ya + 1, nx = xb - xa + 1;
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
    const float* base = logits + (int64_t)b * g.h * g.w * g.ld;
    for (int p = threadIdx.x; p < ny * nx; p += blockDim.x) {
        const int y = ya + p / nx, x = xa + p % nx;
        int y0, y1, x0, x1; float ty, tx;
        src_tap(y, g.sy, g.h, y0, y1, ty);
        src_tap(x, g.sx, g.w, x0, x1, tx);
        float wy = 0.f, wx = 0.f;
        if (y0 == i) wy += 1.f - ty;
        if (y1 == i) wy += ty;
        if (x0 == j) wx += 1.f - tx;
        if (x1 == j) wx += tx;
        const float wgt = wy * wx;
        if (wgt == 0.f) continue;
        const int target = ct.lut[teacher[((int64_t)b * g.H + y) * g.W + x]];
        if (target < 0) continue;
        const float* ptl = base + ((int64_t)y0 * g.w + x0) * g.ld;
        const float* ptr = base + ((int64_t)y0 * g.w + x1) * g.ld;
        const float* pbl = base + ((int64_t)y1 * g.w + x0) * g.ld;
        const float* pbr = base + ((int64_t)y1 * g.w + x1) * g.ld;
        float z[KMAX];
        float zmax = -3.0e38f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < g.K) {
                const int c = ct.idx[k];
                z[k] = bilerp(ptl[c], ptr[c], pbl[c], pbr[c], tx, ty);
                zmax = fmaxf(zmax, z[k]);
            } else z[k] = 0.f;
        }
        float ssum = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < g.K) { z[k] = __expf(z[k] - zmax); ssum += z[k]; }
        const float f = wgt * inv_n, rs = 1.f / ssum;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < g.K) acc[k] += f * (z[k] * rs - (k == target ? 1.f : 0.f));
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) s_part[wave][k] = s;
    }
    __syncthreads();
    float* out = dlogits + (((int64_t)b * g.h + i) * g.w + j) * ldd;
    for (int c = threadIdx.x; c < ldd; c += blockDim.x) {
        float v = 0.f;
        if (c < 256) {
            const int k = ct.lut[c];
            if (k >= 0) v = (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k]);
            if (k >= 0 && !(nvalid > 0.5)) v = empty_val;          // no valid pixel in the batch: 0, or the reference's 0 / 0 (AMS_OPT_NAN_GRADS)
        }
        out[c] = v;
    }
}

int launch_ce_grad(const float* logits, int ld, int B, int h, int w, const int32_t* cls, int K, int H, int W,
                   const uint8_t* teacher, int NC, const double* loss_and_count, float* dlogits, int ldd, hipStream_t st, float empty_val) {
    ClassTable ct;
    int rc = fill_class_table(cls, K, NC, &ct);
    if (rc) return rc;
    AMS_REQUIRE(teacher && loss_and_count && dlogits, "ce_grad: null pointer");
    AMS_REQUIRE(ldd >= NC && ldd <= 256, "ce_grad: ldd=%d must hold %d classes", ldd, NC);
    const HeadGeom g = head_geom(ld, B, h, w, K, H, W, NC);
    const dim3 grid(w, h, B);
    note_kernel(K <= 8 ? "ce_grad_kernel<8>" : K <= 20 ? "ce_grad_kernel<20>" : "ce_grad_kernel<32>");
    if (K <= 8)
        hipLaunchKernelGGL(ce_grad_kernel<8>, grid, dim3(256), 0, st, logits, g, ct, teacher, loss_and_count, dlogits, ldd, empty_val);
    else if (K <= 20)
        hipLaunchKernelGGL(ce_grad_kernel<20>, grid, dim3(256), 0, st, logits, g, ct, teacher, loss_and_count, dlogits, ldd, empty_val);
    else
        hipLaunchKernelGGL(ce_grad_kernel<32>, grid, dim3(256), 0, st, logits, g, ct, teacher, loss_and_count, dlogits, ldd, empty_val);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Loss AND its gradient in one pass over the full-resolution pixels (the fine-tune step: replaces upsample_argmax (loss only) +
// ce_grad, which evaluates every pixel's softmax once per low-resolution cell it touches: four times).
// Block = (CB cell columns, one source-row band, image): the band of source row i0 is every output row y with floor(y * sy) == i0; a
// thread owns one output column and walks down the band with the K horizontally interpolated values of the two source rows in
// registers (as the forward kernel does), evaluates the softmax once per pixel and accumulates (softmax - onehot) weighted by
// (1 - ty) for cell row i0 and by ty for cell row i0 + 1.  The columns are then folded into the block's cell columns in a fixed
// order through LDS (weights 1 - tx / tx); a block also walks the pixel columns of the cell column to its left, so every cell it
// owns is complete in x.  Two partial planes come out — T[b][i0][j] and B[b][i0 + 1][j], each element written exactly once (no
// atomics: run-to-run identical) — and ce_combine_kernel adds them, applies 1 / (valid pixels) and scatters to the class columns.
// ---------------------------------------------------------------------------------------------------------
// cell columns per block: ten, so that the (CB + 1) x 17 + 2 = 189 pixel columns of a block fill its 192 threads and the 8-frame launch
// (7 x 33 x 8 = 1848 blocks) fits ONE round of the 2048 resident blocks; with eight it was 2376 blocks of 155 live threads: two rounds
constexpr int kCeCB = 10;

// SOFT: the target distribution of a pixel is the softmax of the K gathered teacher logits instead of the one-hot row of its label: per-pixel
// loss sum_k p_k (log sum exp(z - max) - (z_k - max)) (TensorFlow's xent kernel), gradient softmax(z) - p; the mask and the mean's denominator
// still come from the hard labels (weights = reduce_sum(filtered_labels_onehot), utils/graph_utils.py:397, 406-408).
template <int KMAX, bool SOFT = false>
__global__ __launch_bounds__(192) void ce_loss_grad_kernel(const float* __restrict__ logits, HeadGeom g, ClassTable ct,
                                                           const uint8_t* __restrict__ teacher, double* __restrict__ loss,
                                                           float* __restrict__ partT, float* __restrict__ partB, SoftTeacher sft) {
    __shared__ float s_val[192][2 * KMAX + 1];
    __shared__ float s_wx[192][2];            // weight of the column towards its left / right source column
    __shared__ int s_x0[192];
    __shared__ double s_loss[3];
    __shared__ int s_cnt[3];
    const int b = blockIdx.z, i0 = blockIdx.y, j_lo = blockIdx.x * kCeCB;
    const int i1 = i0 + 1 < g.h ? i0 + 1 : g.h - 1;
    // pixel rows of the band: floor(y * sy) == i0 (sy = 0: a single source row holds every output row)
    int ybeg, yend;
    {
        const float inv = g.sy > 0.f ? 1.f / g.sy : 0.f;
        int y = g.sy > 0.f ? (int)(i0 * inv) - 2 : 0;
        if (y < 0) y = 0;
        while (y < g.H && (int)floorf(__fmul_rn((float)y, g.sy)) < i0) ++y;
        ybeg = y;
        while (y < g.H && (int)floorf(__fmul_rn((float)y, g.sy)) == i0) ++y;
        yend = y;
    }
    // pixel columns: those whose left source column x0 lies in [j_lo - 1, j_lo + CB - 1]
    int xbeg;
    {
        const float inv = g.sx > 0.f ? 1.f / g.sx : 0.f;
        const int jl = j_lo - 1 < 0 ? 0 : j_lo - 1;
        int x = g.sx > 0.f ? (int)(jl * inv) - 2 : 0;
        if (x < 0) x = 0;
        while (x < g.W && (int)floorf(__fmul_rn((float)x, g.sx)) < jl) ++x;
        xbeg = x;
    }
    const int x = xbeg + threadIdx.x;
    int x0 = 0, x1 = 0; float tx = 0.f;
    bool live = x < g.W;
    if (live) { src_tap(x, g.sx, g.w, x0, x1, tx); live = x0 <= j_lo + kCeCB - 1; }
    float gt[KMAX], gb[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { gt[k] = 0.f; gb[k] = 0.f; }
    double my_loss = 0.0;                     // integer multiples of 2^-20 per pixel: exact, order-independent sums (upsample_argmax_kernel)
    int my_cnt = 0;
    if (live) {
        const float* base = logits + (int64_t)b * g.h * g.w * g.ld;
        const float* ptl = base + ((int64_t)i0 * g.w + x0) * g.ld;
        const float* ptr = base + ((int64_t)i0 * g.w + x1) * g.ld;
        const float* pbl = base + ((int64_t)i1 * g.w + x0) * g.ld;
        const float* pbr = base + ((int64_t)i1 * g.w + x1) * g.ld;
        float top[KMAX], bot[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = ct.idx[k < g.K ? k : 0];
            top[k] = __fadd_rn(ptl[c], __fmul_rn(__fsub_rn(ptr[c], ptl[c]), tx));
            bot[k] = __fadd_rn(pbl[c], __fmul_rn(__fsub_rn(pbr[c], pbl[c]), tx));
        }
        // only columns whose x0 is an OWNED cell column count towards the loss (the overlap column belongs to the block on the left)
        const bool own = x0 >= j_lo;
        // soft targets: the pixel column's taps into the teacher grid are fixed for the thread
        int qx0 = 0, qx1 = 0; float qtx = 0.f;
        const float* qbase = nullptr;
        if constexpr (SOFT) {
            src_tap(x, sft.sx, sft.tw, qx0, qx1, qtx);
            qbase = sft.t + (int64_t)b * sft.th * sft.tw * sft.ld;
        }
        // Eight rows at a time: their label bytes, then their class-table entries, are requested together — the two dependent loads per
        // row were the kernel (63 % of the wave cycles waiting with six waves per SIMD to hide a ~1 us chain per row).  Same arithmetic,
        // same row order: the same bits.
        const uint8_t* tcol = teacher + (int64_t)b * g.H * g.W + x;
        for (int yb = ybeg; yb < yend; yb += 8) {
          int tgt[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) tgt[r] = tcol[(int64_t)(yb + r < yend ? yb + r : yend - 1) * g.W];
#pragma unroll
          for (int r = 0; r < 8; ++r) tgt[r] = ct.lut[tgt[r]];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int y = yb + r;
            const int target = tgt[r];
            if (y >= yend || target < 0) continue;
            const float src = __fmul_rn((float)y, g.sy);
            const float ty = __fsub_rn(src, floorf(src));
            float z[KMAX];
            float zmax = -3.0e38f, zt = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                z[k] = __fadd_rn(top[k], __fmul_rn(__fsub_rn(bot[k], top[k]), ty));
                if (k < g.K) zmax = fmaxf(zmax, z[k]);
                if (k == target) zt = z[k];
            }
            float pt[SOFT ? KMAX : 1];
            float zs[SOFT ? KMAX : 1];                    // z_k - max, kept for the soft loss
            if constexpr (SOFT) {
                int qy0, qy1; float qty;
                src_tap(y, sft.sy, sft.th, qy0, qy1, qty);
                const float* qtl = qbase + ((int64_t)qy0 * sft.tw + qx0) * sft.ld;
                float pmax = -3.0e38f;
                if (qty == 0.f && qtx == 0.f) {           // on a grid point (always, when the teacher logits come at the label size)
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) { pt[k] = qtl[ct.idx[k < g.K ? k : 0]]; if (k < g.K) pmax = fmaxf(pmax, pt[k]); }
                } else {
                    const float* qtr = qbase + ((int64_t)qy0 * sft.tw + qx1) * sft.ld;
                    const float* qbl = qbase + ((int64_t)qy1 * sft.tw + qx0) * sft.ld;
                    const float* qbr = qbase + ((int64_t)qy1 * sft.tw + qx1) * sft.ld;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const int c = ct.idx[k < g.K ? k : 0];
                        pt[k] = bilerp(qtl[c], qtr[c], qbl[c], qbr[c], qtx, qty);
                        if (k < g.K) pmax = fmaxf(pmax, pt[k]);
                    }
                }
                float psum = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < g.K) { pt[k] = __expf(pt[k] - pmax); psum += pt[k]; } else pt[k] = 0.f;
                const float rp = 1.f / psum;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) pt[k] *= rp;
            }
            float ssum = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < g.K) { if constexpr (SOFT) zs[k] = z[k] - zmax; z[k] = __expf(z[k] - zmax); ssum += z[k]; }
            const float rs = 1.f / ssum;
            if constexpr (SOFT) {
                const float lse = __logf(ssum);
                float lp = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < g.K) lp += pt[k] * (lse - zs[k]);
                if (own) { my_loss += rint((double)lp * 1048576.0); my_cnt += 1; }
            } else {
            if (own) { my_loss += rint((double)((zmax + __logf(ssum)) - zt) * 1048576.0); my_cnt += 1; }
            }
            const float wt = 1.f - ty;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < g.K) {
                    float want;
                    if constexpr (SOFT) want = pt[k]; else want = k == target ? 1.f : 0.f;
                    const float d = z[k] * rs - want;
                    gt[k] += wt * d;
                    gb[k] += ty * d;
                }
          }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { s_val[threadIdx.x][k] = gt[k]; s_val[threadIdx.x][KMAX + k] = gb[k]; }
    s_x0[threadIdx.x] = live ? x0 : -2;
    s_wx[threadIdx.x][0] = 1.f - tx;
    s_wx[threadIdx.x][1] = x1 != x0 ? tx : 0.f;       // clamped right neighbour (last source column): tx is 0 there anyway
    my_loss = wave_sum(my_loss);
    my_cnt = (int)wave_sum((float)my_cnt);
    if ((threadIdx.x & 63) == 0) { s_loss[threadIdx.x >> 6] = my_loss; s_cnt[threadIdx.x >> 6] = my_cnt; }
    __syncthreads();
    // fold the pixel columns into the cell columns: entry e = (cell column jj, plane tb, class k), columns in ascending order
    for (int e = threadIdx.x; e < kCeCB * 2 * KMAX; e += blockDim.x) {
        const int jj = e / (2 * KMAX), r = e - jj * (2 * KMAX);
        const int j = j_lo + jj;
        const int tb = r / KMAX, k = r - tb * KMAX;
        if (j >= g.w || k >= g.K) continue;
        float s = 0.f;
        for (int t = 0; t < (int)blockDim.x; ++t) {
            const int c0 = s_x0[t];
            if (c0 == j) s += s_wx[t][0] * s_val[t][r];
            else if (c0 == j - 1) s += s_wx[t][1] * s_val[t][r];
        }
        if (tb == 0) partT[(((int64_t)b * g.h + i0) * g.w + j) * KMAX + k] = s;
        else if (i0 + 1 < g.h) partB[(((int64_t)b * g.h + i0 + 1) * g.w + j) * KMAX + k] = s;
    }
    if (threadIdx.x == 0) {                   // the block's pair (no atomics: ce_loss_sum_kernel adds the pairs)
        double ls = 0; int cn = 0;
        for (int i = 0; i < 3; ++i) { ls += s_loss[i]; cn += s_cnt[i]; }
        double* mine = loss + 2 * (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        mine[0] = ls; mine[1] = (double)cn;
    }
}

// dlogits[b][i][j][c] = (T + B)[b][i][j][k(c)] / valid pixels; class columns outside the subset and the pad columns get 0
template <int KMAX>
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* __restrict__ partT, const float* __restrict__ partB, int64_t cells,
                                                         int w_cells, int h_cells, ClassTable ct, int K,
                                                         const double* __restrict__ loss_and_count, float* __restrict__ dlogits, int ldd,
                                                         float empty_val) {
    const double nvalid = loss_and_count[1];
    const float inv_n = nvalid > 0.5 ? (float)(1.0 / nvalid) : 0.f;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < cells * ldd; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = e / ldd;
        const int c = (int)(e - cell * ldd);
        const int k = c < 256 ? ct.lut[c] : -1;
        float v = 0.f;
        if (k >= 0) {
            const int i = (int)((cell / w_cells) % h_cells);
            const float t = partT[cell * KMAX + k];
            v = (i > 0 ? t + partB[cell * KMAX + k] : t) * inv_n;          // row 0 has no band above it
            if (!(nvalid > 0.5)) v = empty_val;                             // no valid pixel: 0, or the reference's 0 / 0 (AMS_OPT_NAN_GRADS)
        }
        dlogits[e] = v;
    }
}

bool ce_loss_grad_supported(int w, int W) {
    const int per_cell = w > 1 ? (W - 1 + w - 2) / (w - 1) + 1 : W;      // output columns per source column, rounded up, + 1
    return (kCeCB + 1) * per_cell + 2 <= 192;
}

// two gradient planes, then the blocks' (loss, count) pairs as doubles (4 floats per block)
size_t ce_loss_grad_scratch(int B, int h, int w, int K) {
    return (size_t)2 * B * h * w * (K <= 8 ? 8 : K <= 20 ? 20 : 32) + 4 + (size_t)4 * cdiv(w, kCeCB) * h * B;
}

// loss[0] = sum of the blocks' loss sums, loss[1] = sum of their valid-pixel counts: exact (integer multiples of 2^-20 / integers in f64), so the
// order does not matter; replaces 2 x 1848 atomicAdd(double) on ONE pair of addresses, which took 60 of the loss kernel's 73 us at 8 frames
// (device-scope atomics of eight XCDs on one line are serialised at the memory side)
__global__ __launch_bounds__(256) void ce_loss_sum_kernel(const double* __restrict__ blk, int n, double* __restrict__ loss) {
    __shared__ double s0[4], s1[4];
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { a += blk[2 * (int64_t)i]; c += blk[2 * (int64_t)i + 1]; }
    a = wave_sum(a); c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { loss[0] = ((s0[0] + s0[1]) + (s0[2] + s0[3])) * (1.0 / 1048576.0); loss[1] = (s1[0] + s1[1]) + (s1[2] + s1[3]); }
}

// pass 1: loss[0] += CE sum, loss[1] += valid pixels (loss zeroed here), unnormalised gradient planes into scratch
int launch_ce_loss_grad(const float* logits, int ld, int B, int h, int w, const int32_t* cls, int K, int H, int W, const uint8_t* teacher,
                        int NC, double* loss, float* scratch, hipStream_t st, const float* soft_logits, int soft_h, int soft_w) {
    ClassTable ct;
    int rc = fill_class_table(cls, K, NC, &ct);
    if (rc) return rc;
    AMS_REQUIRE(teacher && loss && scratch, "ce_loss_grad: null pointer");
    const HeadGeom g = head_geom(ld, B, h, w, K, H, W, NC);
    // a block's pixel columns: (CB + 1) source columns' worth
    const int per_cell = w > 1 ? (W - 1 + w - 2) / (w - 1) + 1 : W;
    AMS_REQUIRE((kCeCB + 1) * per_cell + 2 <= 192, "ce_loss_grad: %d output columns per source column do not fit a block", per_cell);
    const int KM = K <= 8 ? 8 : K <= 20 ? 20 : 32;
    float* partT = scratch;
    float* partB = scratch + (size_t)B * h * w * KM;
    const dim3 grid(cdiv(w, kCeCB), h, B);
    const size_t planes = (size_t)2 * B * h * w * KM;
    double* blk = reinterpret_cast<double*>(scratch + (planes + 3) / 4 * 4);          // 16-byte aligned behind the planes
    SoftTeacher sft;
    memset(&sft, 0, sizeof(sft));
    if (soft_logits) {
        AMS_REQUIRE(soft_h >= 1 && soft_w >= 1 && soft_h <= H && soft_w <= W, "ce_loss_grad: teacher logits of %d x %d for labels of %d x %d", soft_h, soft_w, H, W);
        sft.t = soft_logits; sft.th = soft_h; sft.tw = soft_w; sft.ld = NC;
        sft.sy = H > 1 ? (float)(soft_h - 1) / (float)(H - 1) : 0.f;          // as head_geom does for the student's own logits
        sft.sx = W > 1 ? (float)(soft_w - 1) / (float)(W - 1) : 0.f;
    }
    note_kernel(soft_logits ? "ce_loss_grad_kernel<soft>" : "ce_loss_grad_kernel");
    if (soft_logits) {
        if (KM == 8) hipLaunchKernelGGL((ce_loss_grad_kernel<8, true>), grid, dim3(192), 0, st, logits, g, ct, teacher, blk, partT, partB, sft);
        else if (KM == 20) hipLaunchKernelGGL((ce_loss_grad_kernel<20, true>), grid, dim3(192), 0, st, logits, g, ct, teacher, blk, partT, partB, sft);
        else hipLaunchKernelGGL((ce_loss_grad_kernel<32, true>), grid, dim3(192), 0, st, logits, g, ct, teacher, blk, partT, partB, sft);
    } else {
    if (KM == 8) hipLaunchKernelGGL((ce_loss_grad_kernel<8, false>), grid, dim3(192), 0, st, logits, g, ct, teacher, blk, partT, partB, sft);
    else if (KM == 20) hipLaunchKernelGGL((ce_loss_grad_kernel<20, false>), grid, dim3(192), 0, st, logits, g, ct, teacher, blk, partT, partB, sft);
    else hipLaunchKernelGGL((ce_loss_grad_kernel<32, false>), grid, dim3(192), 0, st, logits, g, ct, teacher, blk, partT, partB, sft);
    }
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_loss_sum_kernel, dim3(1), dim3(256), 0, st, blk, (int)(grid.x * grid.y * grid.z), loss);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// pass 2 (after the valid-pixel count is final, i.e. after its cross-rank sum in a data-parallel step)
int launch_ce_combine(int B, int h, int w, const int32_t* cls, int K, int NC, const double* loss_and_count, const float* scratch,
                      float* dlogits, int ldd, hipStream_t st, float empty_val) {
    ClassTable ct;
    int rc = fill_class_table(cls, K, NC, &ct);
    if (rc) return rc;
    AMS_REQUIRE(ldd >= NC && ldd <= 256, "ce_combine: ldd=%d must hold %d classes", ldd, NC);
    const int KM = K <= 8 ? 8 : K <= 20 ? 20 : 32;
    const int64_t cells = (int64_t)B * h * w;
    const float* partT = scratch;
    const float* partB = scratch + (size_t)cells * KM;
    const int grid = (int)(cdiv64(cells * ldd, 256) < 2048 ? cdiv64(cells * ldd, 256) : 2048);
    if (KM == 8) hipLaunchKernelGGL(ce_combine_kernel<8>, dim3(grid), dim3(256), 0, st, partT, partB, cells, w, h, ct, K, loss_and_count, dlogits, ldd, empty_val);
    else if (KM == 20) hipLaunchKernelGGL(ce_combine_kernel<20>, dim3(grid), dim3(256), 0, st, partT, partB, cells, w, h, ct, K, loss_and_count, dlogits, ldd, empty_val);
    else hipLaunchKernelGGL(ce_combine_kernel<32>, dim3(grid), dim3(256), 0, st, partT, partB, cells, w, h, ct, K, loss_and_count, dlogits, ldd, empty_val);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// phi-score confusion matrix between two teacher label maps (SemanticNetwork.py:124-139): pixels whose label is in the
// subset in BOTH maps count 1 at [before][after].
__global__ __launch_bounds__(256) void cross_conf_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int64_t n,
                                                         ClassTable ct, int K, unsigned long long* __restrict__ conf) {
    __shared__ int s_conf[kMaxK * kMaxK];
    for (int e = threadIdx.x; e < K * K; e += blockDim.x) s_conf[e] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ka = ct.lut[a[i]], kb = ct.lut[b[i]];
        if (ka >= 0 && kb >= 0) atomicAdd(&s_conf[ka * K + kb], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * K; e += blockDim.x)
        if (s_conf[e]) atomicAdd(&conf[e], (unsigned long long)s_conf[e]);
}

// lut: HOST pointer, 256 entries (teacher id -> subset index or -1)
int launch_cross_confusion(const uint8_t* a, const uint8_t* b, int64_t n, const int32_t* lut, int K, int64_t* conf,
                           hipStream_t st) {
    AMS_REQUIRE(K > 0 && K <= kMaxK, "cross_confusion: K=%d out of range", K);
    ClassTable ct;
    for (int i = 0; i < 256; ++i) ct.lut[i] = lut[i];
    for (int k = 0; k < kMaxK; ++k) ct.idx[k] = 0;
    AMS_CHECK_HIP(hipMemsetAsync(conf, 0, sizeof(int64_t) * K * K, st));
    int grid = (int)cdiv64(n, 256 * 16);
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(cross_conf_kernel, dim3(grid), dim3(256), 0, st, a, b, n, ct, K, (unsigned long long*)conf);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

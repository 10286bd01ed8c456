// Frame ingest on the device (reference run.py:179-183, :415-421): the caller's cv2.resize of a decoded
// frame to [H, 2H] (INTER_LINEAR) with BGR -> RGB, and of the teacher label map (INTER_NEAREST).  With this the uplink
// hands over raw uint8 frames at source resolution (e.g. A2D2 1920x1208: 7 MB) and the 512x1024 network input never
// crosses PCIe as float.  HBM-bound byte work: one thread per output pixel, 3 channels, source rows read through L2.
//
// Arithmetic is OpenCV's own for uint8 (modules/imgproc/src/resize.cpp, generic path; restated for the tests in oracle/cv_resize.py
// and pinned there by hand-derived vectors): source step 1 / (dst / src) in double; per output index f = float((d + 0.5) * step - 0.5),
// s = floor(f), f -= s; columns zero f where the window leaves the image, rows keep it and clamp the taps; 11-bit weights
// cvRound((1.f - f) * 2048) / cvRound(f * 2048) from float products; horizontal pass in int; vertical pass
// ((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2.  An exact 2x down-scale is cv::resize's INTER_AREA shortcut
// (a + b + c + d + 2) >> 2; equal sizes copy.  INTER_NEAREST: s = min(floor(d * step), size - 1).  The host path
// ams_amd/utils.py computes the same bits.  (The file is compiled with -ffp-contract=off: no fused multiply-add may merge the
// float roundings above.)
#include "common.hpp"
#include "kernels.hpp"

namespace ams {

struct IngestGeom { int Hs, Ws, H, W, C, swap_rb, mode2x; double sy, sx; };

// one axis of cv::resize's tap table
__device__ __forceinline__ void fixed_tap(int d, double step, int n_in, bool zero_at_border, int& t0, int& t1, int& w0, int& w1) {
    float f = (float)(((double)d + 0.5) * step - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (zero_at_border && (s < 0 || s >= n_in - 1)) {
        s = s < 0 ? 0 : n_in - 1;
        f = 0.f;
    }
    t0 = s < 0 ? 0 : (s > n_in - 1 ? n_in - 1 : s);
    t1 = s + 1 < 0 ? 0 : (s + 1 > n_in - 1 ? n_in - 1 : s + 1);
    w0 = __float2int_rn((1.f - f) * 2048.f);       // cvRound: nearest, halves to even
    w1 = __float2int_rn(f * 2048.f);
}

template <int MODE>       // 0: nearest, 1: linear (fixed point), 2: exact 2x box average, 3: copy
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, IngestGeom g) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    if (ox >= g.W) return;
    uint8_t* out = dst + ((int64_t)oy * g.W + ox) * g.C;
    if (MODE == 0 || MODE == 3) {
        long sy = oy, sx = ox;
        if (MODE == 0) {
            sy = (long)floor((double)oy * g.sy);
            sx = (long)floor((double)ox * g.sx);
            if (sy > g.Hs - 1) sy = g.Hs - 1;
            if (sx > g.Ws - 1) sx = g.Ws - 1;
        }
        const uint8_t* p = src + ((int64_t)sy * g.Ws + sx) * g.C;
        for (int c = 0; c < g.C; ++c) out[c] = p[g.swap_rb ? g.C - 1 - c : c];
        return;
    }
    if (MODE == 2) {
        const uint8_t* p0 = src + ((int64_t)(2 * oy) * g.Ws + 2 * ox) * g.C;
        const uint8_t* p1 = p0 + (int64_t)g.Ws * g.C;
        for (int c = 0; c < g.C; ++c) {
            const int sc = g.swap_rb ? g.C - 1 - c : c;
            out[c] = (uint8_t)(((int)p0[sc] + (int)p0[g.C + sc] + (int)p1[sc] + (int)p1[g.C + sc] + 2) >> 2);
        }
        return;
    }
    int y0, y1, x0, x1, b0, b1, a0, a1;
    fixed_tap(oy, g.sy, g.Hs, false, y0, y1, b0, b1);
    fixed_tap(ox, g.sx, g.Ws, true, x0, x1, a0, a1);
    const uint8_t* p00 = src + ((int64_t)y0 * g.Ws + x0) * g.C;
    const uint8_t* p01 = src + ((int64_t)y0 * g.Ws + x1) * g.C;
    const uint8_t* p10 = src + ((int64_t)y1 * g.Ws + x0) * g.C;
    const uint8_t* p11 = src + ((int64_t)y1 * g.Ws + x1) * g.C;
    for (int c = 0; c < g.C; ++c) {
        const int sc = g.swap_rb ? g.C - 1 - c : c;
        const int d0 = (int)p00[sc] * a0 + (int)p01[sc] * a1;
        const int d1 = (int)p10[sc] * a0 + (int)p11[sc] * a1;
        const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
        out[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

int launch_resize_u8(const uint8_t* src, int Hs, int Ws, int C, int mode, int swap_rb, uint8_t* dst, int H, int W, hipStream_t st) {
    AMS_REQUIRE(src && dst && Hs > 0 && Ws > 0 && H > 0 && W > 0, "resize: bad geometry %dx%d -> %dx%d", Hs, Ws, H, W);
    AMS_REQUIRE(C >= 1 && C <= 4 && (mode == 0 || mode == 1), "resize: C=%d mode=%d", C, mode);
    AMS_REQUIRE(!swap_rb || C == 3, "resize: channel swap needs 3 channels");
    IngestGeom g;
    g.Hs = Hs; g.Ws = Ws; g.H = H; g.W = W; g.C = C; g.swap_rb = swap_rb; g.mode2x = 0;
    g.sy = 1.0 / ((double)H / (double)Hs);          // cv::resize: inv_scale = dsize / ssize; scale = 1. / inv_scale
    g.sx = 1.0 / ((double)W / (double)Ws);
    const dim3 grid(cdiv(W, 256), H);
    int k = mode;
    if (mode == 1 && Hs == H && Ws == W) k = 3;
    else if (mode == 1 && Hs == 2 * H && Ws == 2 * W) k = 2;
    static const char* names[4] = {"resize_u8_kernel<0>", "resize_u8_kernel<1>", "resize_u8_kernel<2>", "resize_u8_kernel<3>"};
    note_kernel(names[k]);
    if (k == 0) hipLaunchKernelGGL(resize_u8_kernel<0>, grid, dim3(256), 0, st, src, dst, g);
    else if (k == 1) hipLaunchKernelGGL(resize_u8_kernel<1>, grid, dim3(256), 0, st, src, dst, g);
    else if (k == 2) hipLaunchKernelGGL(resize_u8_kernel<2>, grid, dim3(256), 0, st, src, dst, g);
    else hipLaunchKernelGGL(resize_u8_kernel<3>, grid, dim3(256), 0, st, src, dst, g);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

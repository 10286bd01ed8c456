// Frame ingest on the device (reference run.py:179-183, :415-421): the caller's cv2.resize of a decoded
// frame to [H, 2H] (INTER_LINEAR) with BGR -> RGB, and of the teacher label map (INTER_NEAREST).  With this the uplink
// hands over raw uint8 frames at source resolution (e.g. A2D2 1920x1208: 7 MB) and the 512x1024 network input never
// crosses PCIe as float.  HBM-bound byte work: one thread per output pixel, 3 channels, source rows read through L2.
//
// Arithmetic follows ams_amd/utils.py resize_linear / resize_nearest operation by operation in f64 (half-pixel centres,
// edge clamp, two lerps, round-half-even), so the device result is bit-identical to that host restatement — which itself
// is within 1 LSB of OpenCV's 11-bit fixed-point uint8 path (cv2 is not available here: unpinned).
#include "common.hpp"
#include "kernels.hpp"

namespace ams {

struct IngestGeom { int Hs, Ws, H, W, C, swap_rb; double ry, rx; };

__device__ __forceinline__ void lin_tap(int o, double ratio, int n_in, int& lo_c, int& hi_c, double& frac) {
    const double pos = ((double)o + 0.5) * ratio - 0.5;
    const double fl = floor(pos);
    frac = pos - fl;
    const long lo = (long)fl;
    lo_c = lo < 0 ? 0 : (lo > n_in - 1 ? n_in - 1 : (int)lo);
    const long hi = lo + 1;
    hi_c = hi < 0 ? 0 : (hi > n_in - 1 ? n_in - 1 : (int)hi);
}

template <int MODE>       // 0: nearest, 1: linear
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, IngestGeom g) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    if (ox >= g.W) return;
    uint8_t* out = dst + ((int64_t)oy * g.W + ox) * g.C;
    if (MODE == 0) {
        long sy = (long)((double)oy * g.ry), sx = (long)((double)ox * g.rx);
        if (sy > g.Hs - 1) sy = g.Hs - 1;
        if (sx > g.Ws - 1) sx = g.Ws - 1;
        const uint8_t* p = src + ((int64_t)sy * g.Ws + sx) * g.C;
        for (int c = 0; c < g.C; ++c) out[c] = p[g.swap_rb ? g.C - 1 - c : c];
        return;
    }
    int y0, y1, x0, x1;
    double fy, fx;
    lin_tap(oy, g.ry, g.Hs, y0, y1, fy);
    lin_tap(ox, g.rx, g.Ws, x0, x1, fx);
    const uint8_t* p00 = src + ((int64_t)y0 * g.Ws + x0) * g.C;
    const uint8_t* p01 = src + ((int64_t)y0 * g.Ws + x1) * g.C;
    const uint8_t* p10 = src + ((int64_t)y1 * g.Ws + x0) * g.C;
    const uint8_t* p11 = src + ((int64_t)y1 * g.Ws + x1) * g.C;
    const double gx = 1.0 - fx, gy = 1.0 - fy;
    for (int c = 0; c < g.C; ++c) {
        const int sc = g.swap_rb ? g.C - 1 - c : c;
        const double top = (double)p00[sc] * gx + (double)p01[sc] * fx;
        const double bot = (double)p10[sc] * gx + (double)p11[sc] * fx;
        double v = rint(top * gy + bot * fy);
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        out[c] = (uint8_t)v;
    }
}

int launch_resize_u8(const uint8_t* src, int Hs, int Ws, int C, int mode, int swap_rb, uint8_t* dst, int H, int W, hipStream_t st) {
    AMS_REQUIRE(src && dst && Hs > 0 && Ws > 0 && H > 0 && W > 0, "resize: bad geometry %dx%d -> %dx%d", Hs, Ws, H, W);
    AMS_REQUIRE(C >= 1 && C <= 4 && (mode == 0 || mode == 1), "resize: C=%d mode=%d", C, mode);
    AMS_REQUIRE(!swap_rb || C == 3, "resize: channel swap needs 3 channels");
    IngestGeom g;
    g.Hs = Hs; g.Ws = Ws; g.H = H; g.W = W; g.C = C; g.swap_rb = swap_rb;
    g.ry = (double)Hs / (double)H;
    g.rx = (double)Ws / (double)W;
    const dim3 grid(cdiv(W, 256), H);
    note_kernel(mode ? "resize_u8_kernel<1>" : "resize_u8_kernel<0>");
    if (mode) hipLaunchKernelGGL(resize_u8_kernel<1>, grid, dim3(256), 0, st, src, dst, g);
    else hipLaunchKernelGGL(resize_u8_kernel<0>, grid, dim3(256), 0, st, src, dst, g);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams

// Probe (round 4): the three-part split GEMM with its operands brought in by LDS-direct loads (global_load_lds_dwordx4: no registers in flight)
// through a ring of D stages, ONE block of four waves per CU, against the library kernel's register ring of two stages at three waves per SIMD.
// Fixed form: y [M,N] = x [M,K] . w [K,N] with w given as three bf16 part panels [3][N][K] (hi, mid, lo); tile 128 rows x 80 columns, 32 k per stage.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I ams_amd/csrc -I include tools/probes/gemm_async_probe.hip -o gpurun_out/gemm_async_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <utility>
#include "common.hpp"
#include "split_bf16.hpp"
using namespace ams;

constexpr int NT = 5, RM = 2;
constexpr int X_STAGE = 4 * RM * 2 * 1024, W_STAGE = NT * 3 * 1024, STAGE = X_STAGE + W_STAGE;      // bytes

typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
__device__ __forceinline__ void lds_async16(const void* gptr, lds_ptr base, unsigned byte_off) {
    __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void*)(base + byte_off), 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }
#ifdef GAP_CXX_READS
__device__ __forceinline__ u32x4 lds_read16(unsigned byte_off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((__attribute__((address_space(3))) unsigned char*)(size_t)byte_off);
}
#else
__device__ __forceinline__ u32x4 lds_read16(unsigned byte_off) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_off));
    return v;
}
#endif
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int D>
__global__ __launch_bounds__(256, 1) void gemm_async(const float* __restrict__ x, int M, int K, const unsigned short* __restrict__ wp, int N, float* __restrict__ y, int getenv_dbg, unsigned* __restrict__ dump = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.y * (16 * NT);
    const int64_t m_w = (int64_t)blockIdx.x * 128 + wave * 32;
    const int n_stages = K / 32;
    // this lane's global sources: x pieces (r, j) and the weight pieces of instructions u = wave, wave + 4, wave + 8, wave + 12 (u = t * 3 + p)
    const float* xsrc[RM][2];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        int64_t m = m_w + r * 16 + l15;
        if (m > M - 1) m = M - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) xsrc[r][j] = x + m * K + 8 * q + 4 * j;
    }
    const unsigned short* wsrc[4];
    unsigned wdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int u = wave + 4 * i;
        if (u > NT * 3 - 1) u = NT * 3 - 1;               // wave 3's fourth instruction repeats the last piece set (same data, same place)
        const int t = u / 3, p = u - t * 3;
        int n = n0 + 16 * t + l15;
        if (n > N - 1) n = N - 1;
        wsrc[i] = wp + (int64_t)p * N * K + (int64_t)n * K + 8 * q;
        wdst[i] = X_STAGE + u * 1024;
    }
    auto issue = [&](int s) {
        const int sc = s < n_stages ? s : n_stages - 1;
        const unsigned base = (unsigned)(s % D) * STAGE;
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) lds_async16(xsrc[r][j] + sc * 32, (lds_ptr)smem, base + wave * (RM * 2 * 1024) + (r * 2 + j) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_async16(wsrc[i] + sc * 32, (lds_ptr)smem, base + wdst[i]);
    };
    f32x4 acc[RM][NT];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Software pipeline (GAP_PIPE): the fragments of stage s + 1 are read from LDS into a second register set while the MFMAs of stage s run;
    // a slot of the ring is free again as soon as every wave has its fragments in registers.
    auto read_frags = [&](int s, u32x4 (&xr)[RM][2], u32x4 (&wq)[NT][3]) {
        const unsigned base = lds0 + (unsigned)(s % D) * STAGE;
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) xr[r][j] = lds_read16(base + wave * (RM * 2 * 1024) + (r * 2 + j) * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) wq[t][p] = lds_read16(base + X_STAGE + (t * 3 + p) * 1024 + lane * 16);
    };
    auto pin = [&](u32x4 (&xr)[RM][2], u32x4 (&wq)[NT][3]) {
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(xr[r][j]));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(wq[t][p]));
    };
    auto compute = [&](const u32x4 (&xr)[RM][2], const u32x4 (&wq)[NT][3]) {
        bf16x8 x0[RM], x1[RM], x2[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const f32x4 uf = __builtin_bit_cast(f32x4, xr[r][0]), vf = __builtin_bit_cast(f32x4, xr[r][1]);
            const float4 u = make_float4(uf[0], uf[1], uf[2], uf[3]), v = make_float4(vf[0], vf[1], vf[2], vf[3]);
            split8(u, v, x0[r], x1[r], x2[r]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 q0 = __builtin_bit_cast(bf16x8, wq[t][0]), q1 = __builtin_bit_cast(bf16x8, wq[t][1]), q2 = __builtin_bit_cast(bf16x8, wq[t][2]);
#define TERM(QA, XB) _Pragma("unroll") for (int r = 0; r < RM; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QA, XB[r], acc[r][t], 0, 0, 0);
            TERM(q2, x0) TERM(q0, x2) TERM(q1, x1) TERM(q1, x0) TERM(q0, x1) TERM(q0, x0)
#undef TERM
        }
    };
    if (getenv_dbg & 8) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
        u32x4 xa[RM][2], wa[NT][3], xb[RM][2], wb[NT][3];
        wait_vm<8 * (D - 1)>();
        __builtin_amdgcn_s_barrier();
        read_frags(0, xa, wa);
        wait_lgkm0();
        pin(xa, wa);
        for (int s = 0; s < n_stages; s += 2) {
            // even stage: compute on (xa, wa), fetch stage s + 1 into (xb, wb)
            wait_vm<8 * (D - 2)>();
            __builtin_amdgcn_s_barrier();
            issue(s + D);
            read_frags(s + 1, xb, wb);
            compute(xa, wa);
            wait_lgkm0();
            pin(xb, wb);
            if (s + 1 >= n_stages) break;
            wait_vm<8 * (D - 2)>();
            __builtin_amdgcn_s_barrier();
            issue(s + 1 + D);
            read_frags(s + 2, xa, wa);
            compute(xb, wb);
            wait_lgkm0();
            pin(xa, wa);
        }
    } else {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(d);
    for (int s = 0; s < n_stages; ++s) {
        wait_vm<8 * (D - 2)>();
        __builtin_amdgcn_s_barrier();                      // (no fence: __syncthreads() makes hipcc drain every LDS-direct load in flight)
        issue(s + D - 1);
        u32x4 xr[RM][2], wq[NT][3];
        read_frags(s, xr, wq);
        wait_lgkm0();
        pin(xr, wq);
        compute(xr, wq);
    }
    }
    if (dump && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {
        unsigned* o = dump + STAGE / 4 + 64 * 12 + lane * 4;
        o[0] = __builtin_bit_cast(unsigned, acc[0][0][0]); o[1] = __builtin_bit_cast(unsigned, acc[0][0][1]);
        o[2] = __builtin_bit_cast(unsigned, acc[0][0][2]); o[3] = __builtin_bit_cast(unsigned, acc[0][0][3]);
    }
    // plain stores: lane holds columns n0 + 16 t + 4 q .. + 3 of row m_w + 16 r + l15
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int64_t m = m_w + r * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + 16 * t + 4 * q;
            if (n < N) *reinterpret_cast<float4*>(y + m * N + n) = make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
        }
    }
}

static unsigned short f2bf(float f) {          // round to nearest even
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 68640, K = argc > 2 ? atoi(argv[2]) : 960, N = argc > 3 ? atoi(argv[3]) : 160;
    std::vector<float> hx((size_t)M * K), hw((size_t)K * N);
    srand(1);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / sqrtf((float)K);
    if (getenv("GAP_PAT")) {
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hx[(size_t)m * K + k] = k == 5 ? 1.f + m : 0.f;
        for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) hw[(size_t)k * N + n] = (float)(k * 100 + n);
    }
    std::vector<unsigned short> hp((size_t)3 * N * K);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float w = hw[(size_t)k * N + n];
            const unsigned short h = f2bf(w); const float r1 = w - bf2f(h);
            const unsigned short m = f2bf(r1); const float r2 = r1 - bf2f(m);
            hp[(size_t)0 * N * K + (size_t)n * K + k] = h; hp[(size_t)1 * N * K + (size_t)n * K + k] = m; hp[(size_t)2 * N * K + (size_t)n * K + k] = f2bf(r2);
        }
    float *dx, *dy; unsigned short* dp;
    (void)hipMalloc(&dx, hx.size() * 4); (void)hipMalloc(&dy, (size_t)M * N * 4); (void)hipMalloc(&dp, hp.size() * 2);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 grid((M + 127) / 128, (N + 16 * NT - 1) / (16 * NT));
    const int dbg = getenv("GAP_DBG") ? atoi(getenv("GAP_DBG")) : 0;
    auto run = [&](int D) {
        const size_t lds = (size_t)D * STAGE;
        auto go = [&] {
            if (D == 2) { (void)hipFuncSetAttribute((const void*)gemm_async<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(gemm_async<2>, grid, dim3(256), lds, 0, dx, M, K, dp, N, dy, dbg); }
            else if (D == 3) { (void)hipFuncSetAttribute((const void*)gemm_async<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(gemm_async<3>, grid, dim3(256), lds, 0, dx, M, K, dp, N, dy, dbg); }
            else if (D == 4) { (void)hipFuncSetAttribute((const void*)gemm_async<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(gemm_async<4>, grid, dim3(256), lds, 0, dx, M, K, dp, N, dy, dbg); }
            else { (void)hipFuncSetAttribute((const void*)gemm_async<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(gemm_async<5>, grid, dim3(256), lds, 0, dx, M, K, dp, N, dy, dbg); }
        };
        for (int i = 0; i < 3; ++i) go();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) go();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // check a few rows against f64
        std::vector<float> hy((size_t)M * N);
        (void)hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        if (getenv("GAP_DEBUG")) {
            for (int m : {0, 1, 16, 33})
                for (int n : {0, 1, 4, 16, 79}) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)k * N + n];
                    printf("  y[%d][%d] = %g, want %g\n", m, n, hy[(size_t)m * N + n], s);
                }
        }
        for (int m : {0, 1, 17, 127, 128 % M, 5000 % M, M - 1})
            for (int n = 0; n < N; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)k * N + n];
                worst = fmax(worst, fabs(s - hy[(size_t)m * N + n])); scale = fmax(scale, fabs(s));
            }
        printf("D=%d  lds %zu KB  %.1f us  (%.0f GB/s algorithmic, %.0f TFLOP/s f32-equivalent)  max err %.2e of %.2f  hipErr %d\n", D, lds / 1024, ms * 100,
               4.0 * ((double)M * (K + N) + (double)K * N) / (ms * 1e-4) / 1e9, 2.0 * M * K * N / (ms * 1e-4) / 1e12, worst, scale, (int)hipGetLastError());
    };
    if (getenv("GAP_DUMP")) {
        unsigned* dd; (void)hipMalloc(&dd, STAGE + 64 * 16 * 4);
        (void)hipFuncSetAttribute((const void*)gemm_async<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        hipLaunchKernelGGL(gemm_async<2>, grid, dim3(256), 2 * STAGE, 0, dx, M, K, dp, N, dy, 0, dd);
        std::vector<unsigned> hd(STAGE / 4 + 64 * 16);
        (void)hipMemcpy(hd.data(), dd, STAGE + 64 * 16 * 4, hipMemcpyDeviceToHost);
        for (int lane : {0, 1, 16, 17}) {
            printf("  lane %2d acc[0][0]:", lane);
            for (int e = 0; e < 4; ++e) { float f; memcpy(&f, &hd[STAGE / 4 + 64 * 12 + lane * 4 + e], 4); printf(" %g", f); }
            printf("\n");
        }
        for (int lane : {0, 1, 16, 17, 63}) {
            printf("  lane %2d x frag:", lane);
            for (int e = 0; e < 8; ++e) { float f; memcpy(&f, &hd[STAGE / 4 + lane * 12 + e], 4); printf(" %g", f); }
            printf("   want x[%d][%d..]:", lane & 15, 8 * (lane >> 4));
            for (int e = 0; e < 8; ++e) printf(" %g", hx[(size_t)(lane & 15) * K + 8 * (lane >> 4) + e]);
            printf("\n           w frag (hi):");
            for (int e = 0; e < 8; ++e) { const unsigned u = hd[STAGE / 4 + lane * 12 + 8 + e / 2]; printf(" %g", bf2f((unsigned short)(e & 1 ? u >> 16 : u & 0xffff))); }
            printf("   want w[%d..][%d]:", 8 * (lane >> 4), lane & 15);
            for (int e = 0; e < 8; ++e) printf(" %g", bf2f(hp[(size_t)(lane & 15) * K + 8 * (lane >> 4) + e]));
            printf("\n");
        }
        int bad_x = 0, bad_w = 0;
        for (int wave = 0; wave < 4; ++wave) for (int r = 0; r < RM; ++r) for (int j = 0; j < 2; ++j) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
            const int m = wave * 32 + r * 16 + (lane & 15), k = 8 * (lane >> 4) + 4 * j + e;
            float f; unsigned u = hd[(wave * 4096 + (r * 2 + j) * 1024 + lane * 16) / 4 + e]; memcpy(&f, &u, 4);
            if (f != hx[(size_t)m * K + k]) { if (bad_x < 5) printf("  x wave %d r %d j %d lane %d e %d: %g want %g\n", wave, r, j, lane, e, f, hx[(size_t)m * K + k]); ++bad_x; }
        }
        for (int t = 0; t < NT; ++t) for (int p = 0; p < 3; ++p) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
            const int n = 16 * t + (lane & 15), k = 8 * (lane >> 4) + e;
            const unsigned u = hd[(X_STAGE + (t * 3 + p) * 1024 + lane * 16) / 4 + e / 2];
            const unsigned short got = (unsigned short)(e & 1 ? u >> 16 : u & 0xffff), want = hp[(size_t)p * N * K + (size_t)n * K + k];
            if (got != want) { if (bad_w < 5) printf("  w t %d p %d lane %d e %d: %04x want %04x\n", t, p, lane, e, got, want); ++bad_w; }
        }
        {
            std::vector<float> hy2((size_t)M * N);
            (void)hipMemcpy(hy2.data(), dy, hy2.size() * 4, hipMemcpyDeviceToHost);
            if (K == 32 && !getenv("GAP_PAT")) {
                // which k contribute?  least squares for c_k in y = sum_k c_k x[m][k] w[k][n]
                std::vector<double> A(32 * 32, 0.0), b(32, 0.0);
                for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
                    double f[32];
                    for (int k = 0; k < 32; ++k) f[k] = (double)hx[(size_t)m * K + k] * hw[(size_t)k * N + n];
                    for (int i = 0; i < 32; ++i) { b[i] += f[i] * hy2[(size_t)m * N + n]; for (int j = 0; j < 32; ++j) A[i * 32 + j] += f[i] * f[j]; }
                }
                // Gaussian elimination
                for (int i = 0; i < 32; ++i) {
                    int piv = i; for (int r2 = i + 1; r2 < 32; ++r2) if (fabs(A[r2 * 32 + i]) > fabs(A[piv * 32 + i])) piv = r2;
                    for (int j = 0; j < 32; ++j) std::swap(A[i * 32 + j], A[piv * 32 + j]); std::swap(b[i], b[piv]);
                    for (int r2 = i + 1; r2 < 32; ++r2) { const double fct = A[r2 * 32 + i] / A[i * 32 + i]; for (int j = i; j < 32; ++j) A[r2 * 32 + j] -= fct * A[i * 32 + j]; b[r2] -= fct * b[i]; }
                }
                double c[32];
                for (int i = 31; i >= 0; --i) { double s2 = b[i]; for (int j = i + 1; j < 32; ++j) s2 -= A[i * 32 + j] * c[j]; c[i] = s2 / A[i * 32 + i]; }
                printf("  coefficient of each k in the result:");
                for (int k = 0; k < 32; ++k) printf(" %.2f", c[k]);
                printf("\n");
            }
            int nz = 0;
            for (size_t i = 0; i < hy2.size(); ++i) if (hy2[i] != 0.f) { if (nz < 6) printf("  nonzero y[%zu][%zu] = %g\n", i / N, i % N, hy2[i]); ++nz; }
            printf("  %d nonzero of %zu\n", nz, hy2.size());
            printf("  after the dump launch: y[0][0..3] = %g %g %g %g, y[1][0] = %g, y[17][5] = %g (want %g)\n", hy2[0], hy2[1], hy2[2], hy2[3], hy2[N], hy2[17 * N + 5],
                   (double)hx[17 * K + 5] * hw[5 * N + 5]);
        }
        printf("LDS dump of stage 0: %d wrong x values, %d wrong w values\n", bad_x, bad_w);
    }
    for (int D : {2, 3, 4, 5}) run(D);
    return 0;
}

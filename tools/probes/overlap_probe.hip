// Does VALU work overlap with bf16 MFMA work on one SIMD of gfx950?  One workgroup per CU; per loop iteration a wave issues
//   M: 64 MFMAs (v_mfma_f32_16x16x32_bf16, or 32 x v_mfma_f32_32x32x16_bf16 = the same FLOPs)      V: 256 independent-enough v_fma_f32
// modes: 0 = M only, 1 = V only, 2 = M then V (blocked), 3 = interleaved 1 MFMA : 4 (8) VALU, 4 = waves 0-3 M only and waves 4-7 V only
// (512 threads: two waves per SIMD).  build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2))
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(p1), "v"(p2))

// the V part as packed f32: 128 v_pk_fma_f32 per wave-iteration = the same 256 FMAs per lane
template <int MODE>
__global__ __launch_bounds__(512) void probe_pk(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x2 w[16];
    for (int i = 0; i < 16; ++i) w[i] = (f32x2){threadIdx.x * 0.25f + i, threadIdx.x * 0.5f + i};
    const f32x2 p1 = {1.0001f, 1.0002f}, p2 = {0.5f, 0.25f};
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 4 && ((wave >> 2) & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 4 && ((wave >> 2) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 7], 0, 0, 0);
                PKFMA((2 * k) & 15); PKFMA((2 * k + 1) & 15);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int k = 0; k < 64; ++k) acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 7], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int k = 0; k < 128; ++k) PKFMA(k & 15);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += w[i][0] + w[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool BIG>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    f32x4 acc[8];
    f32x16 big[4];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.25f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 4 && ((wave >> 2) & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 4 && ((wave >> 2) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
            if (BIG) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    big[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[k & 3], 0, 0, 0);
                    FMA((8 * k) & 15); FMA((8 * k + 1) & 15); FMA((8 * k + 2) & 15); FMA((8 * k + 3) & 15);
                    FMA((8 * k + 4) & 15); FMA((8 * k + 5) & 15); FMA((8 * k + 6) & 15); FMA((8 * k + 7) & 15);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 7], 0, 0, 0);
                    FMA((4 * k) & 15); FMA((4 * k + 1) & 15); FMA((4 * k + 2) & 15); FMA((4 * k + 3) & 15);
                }
            }
        } else {
            if (do_m) {
                if (BIG) {
#pragma unroll
                    for (int k = 0; k < 32; ++k) big[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[k & 3], 0, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 64; ++k) acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 7], 0, 0, 0);
                }
            }
            if (do_v) {
#pragma unroll
                for (int k = 0; k < 256; ++k) FMA(k & 15);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += big[i][j];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool BIG>
static void run(const char* name, float* out, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MODE, BIG>), dim3(256), dim3(threads), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, BIG>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-6s %-52s waves/SIMD %d: %7.1f ns per iteration\n", BIG ? "32x32" : "16x16", name, threads / 256, ms * 1e6 / iters);
}

template <int MODE>
static void run_pk(const char* name, float* out, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL((probe_pk<MODE>), dim3(256), dim3(threads), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe_pk<MODE>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-6s %-52s waves/SIMD %d: %7.1f ns per iteration\n", "pk", name, threads / 256, ms * 1e6 / iters);
}

template <bool BIG>
static void all(float* out) {
    for (int threads : {256, 512}) {
        run<0, BIG>("M only (64 MFMA-equivalents per wave-iteration)", out, threads);
        run<1, BIG>("V only (256 v_fma per wave-iteration)", out, threads);
        run<2, BIG>("M then V, same wave", out, threads);
        run<3, BIG>("M and V interleaved, same wave", out, threads);
        if (threads == 512) run<4, BIG>("waves 0-3 M, waves 4-7 V (one of each per SIMD)", out, threads);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    all<false>(out);
    all<true>(out);
    for (int threads : {256, 512}) {
        run_pk<1>("V only (128 v_pk_fma_f32 = 256 FMAs per lane)", out, threads);
        run_pk<2>("M then V(pk), same wave", out, threads);
        run_pk<3>("M and V(pk) interleaved 1:2, same wave", out, threads);
        if (threads == 512) run_pk<4>("waves 0-3 M, waves 4-7 V(pk)", out, threads);
    }
    return 0;
}

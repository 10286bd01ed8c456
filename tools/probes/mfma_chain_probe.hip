// Dependent-accumulator distance of v_mfma_f32_16x16x32_bf16 (and 16x16x4_f32): one wave per SIMD issues MFMAs whose accumulator
// repeats every D instructions; ns per MFMA.  build: hipcc --offload-arch=gfx950 -O3 mfma_chain_probe.hip -o mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x16_bf16 (K = 16, the pre-gfx950 shape): rate on gfx950
__global__ __launch_bounds__(256) void probe_k16(float* out, int iters) {
    s16x4 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) acc[k % 4] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[k % 4], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int D, bool F32>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    const float fa = threadIdx.x * 0.01f, fb = 0.5f;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (F32) acc[k % D] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[k % D], 0, 0, 0);
            else acc[k % D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k % D], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int D, bool F32>
static void run(float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<D, F32>), dim3(256), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<D, F32>), dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s accumulator reused every %d MFMAs: %6.2f ns per MFMA\n", F32 ? "16x16x4_f32  " : "16x16x32_bf16", D, ms * 1e6 / iters / 64);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(probe_k16, dim3(256), dim3(256), 0, 0, out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe_k16, dim3(256), dim3(256), 0, 0, out, 2000);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x16_bf16 (K = 16), four accumulators: %6.2f ns per MFMA\n", ms * 1e6 / 2000 / 64);
    }
    run<1, false>(out); run<2, false>(out); run<3, false>(out); run<4, false>(out); run<8, false>(out);
    run<1, true>(out); run<2, true>(out); run<4, true>(out); run<8, true>(out);
    return 0;
}

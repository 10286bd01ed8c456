#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// Probe: global -> LDS without registers (gfx950: global_load_lds_dwordx4), DEPTH stages of 4 KB per wave in flight; each stage is then read
// from LDS and summed (so the data really arrives).  Compares with the same stream through registers (one stage ahead).
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_lds(const float* __restrict__ x, float* __restrict__ out, long n_stage_total) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* my = smem + wave * DEPTH * 1024;          // DEPTH stages x 64 lanes x 16 B = 4 KB per stage per wave
    const long wstride = (long)gridDim.x * 4;
    long s = (long)blockIdx.x * 4 + wave;
    float4 acc = make_float4(0, 0, 0, 0);
    // prologue
    long sp = s;
    for (int d = 0; d < DEPTH - 1; ++d, sp += wstride) {
        const long ss = sp < n_stage_total ? sp : n_stage_total - 1;
        __builtin_amdgcn_global_load_lds(x + ss * 256 + lane * 4, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, 0, 0);
    }
    int d_rd = 0, d_wr = DEPTH - 1;
    for (; s < n_stage_total; s += wstride, sp += wstride) {
        const long ss = sp < n_stage_total ? sp : n_stage_total - 1;
        __builtin_amdgcn_global_load_lds(x + ss * 256 + lane * 4, (__attribute__((address_space(3))) void*)(my + d_wr * 1024), 16, 0, 0);
        // wait until only DEPTH-1 loads are outstanding: the oldest stage has landed.  The LDS read is inline asm: a C++ read makes hipcc wait
        // for ALL outstanding LDS-direct loads (vmcnt(0)) because it may alias any of them
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f vv;
        const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(my + d_rd * 1024 + lane * 4);
        asm volatile("s_waitcnt vmcnt(%1)\n\tds_read_b128 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(vv) : "n"(DEPTH - 1), "v"(lds_addr));
        const float4 v = make_float4(vv.x, vv.y, vv.z, vv.w);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        d_rd = d_rd + 1 == DEPTH ? 0 : d_rd + 1;
        d_wr = d_wr + 1 == DEPTH ? 0 : d_wr + 1;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void stream_reg(const float* __restrict__ x, float* __restrict__ out, long n_stage_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wstride = (long)gridDim.x * 4;
    long s = (long)blockIdx.x * 4 + wave;
    float4 acc = make_float4(0, 0, 0, 0);
    float4 cur = *reinterpret_cast<const float4*>(x + (s < n_stage_total ? s : n_stage_total - 1) * 256 + lane * 4);
    for (; s < n_stage_total; s += wstride) {
        const long sn = s + wstride < n_stage_total ? s + wstride : n_stage_total - 1;
        const float4 nxt = *reinterpret_cast<const float4*>(x + sn * 256 + lane * 4);
        acc.x += cur.x; acc.y += cur.y; acc.z += cur.z; acc.w += cur.w;
        cur = nxt;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
int main() {
    const long bytes = 1L << 30;
    float *x, *out;
    hipMalloc(&x, bytes); hipMalloc(&out, 64);
    hipMemset(x, 0, bytes);
    const long n_stage = bytes / 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto fn, const char* name) {
        for (int i = 0; i < 2; ++i) fn();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.1f us  %.0f GB/s\n", name, ms * 200, bytes / (ms / 5 * 1e-3) / 1e9);
    };
    for (int per_cu : {1, 2, 4}) {
        const int blocks = 256 * per_cu;
        printf("-- %d block(s) of 4 waves per CU\n", per_cu);
        time([&] { hipLaunchKernelGGL(stream_reg, dim3(blocks), dim3(256), 0, 0, x, out, n_stage); }, "registers, 1 ahead");
        time([&] { hipLaunchKernelGGL(stream_lds<2>, dim3(blocks), dim3(256), 4 * 2 * 4096, 0, x, out, n_stage); }, "lds async depth 2");
        time([&] { hipLaunchKernelGGL(stream_lds<4>, dim3(blocks), dim3(256), 4 * 4 * 4096, 0, x, out, n_stage); }, "lds async depth 4");
        time([&] { hipLaunchKernelGGL(stream_lds<8>, dim3(blocks), dim3(256), 4 * 8 * 4096, 0, x, out, n_stage); }, "lds async depth 8");
    }
    return 0;
}

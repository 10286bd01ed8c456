// Where does global_load_lds_dwordx4 put lane i's 16 bytes?  Lane i fetches the four floats 4i .. 4i+3; the LDS region is then dumped linearly.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) smem[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds(x + lane * 4, (__attribute__((address_space(3))) void*)smem, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = smem[i];
}
int main() {
    float h[256], *dx, *dout, o[512];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    (void)hipMalloc(&dx, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    (void)hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, dx, dout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int i = 0; i < 40; ++i) printf("%g ", o[i]);
    printf("\n... [64..71]: ");
    for (int i = 64; i < 72; ++i) printf("%g ", o[i]);
    printf("\n... [252..259]: ");
    for (int i = 252; i < 260; ++i) printf("%g ", o[i]);
    printf("\n");
    return 0;
}

// Probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used by the split-bf16 weight-gradient kernel).
// LDS tile T[64 rows][64 cols] of u16 with T[r][c] = r * 64 + c.  Each 16-lane group q reads the 4 x 16 block with rows
// 8q..8q+3 and columns 0..15; lane p supplies the address of T[8q + p/4][4 * (p%4)].  Expected: lane p receives
// T[8q + j][p], j = 0..3.  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o tools/probes/tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, p = lane & 15, q = lane >> 4;
    const unsigned short* addr = &lds[(8 * q + p / 4) * 64 + 4 * (p % 4)];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256];
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
            const int p = lane & 15, q = lane >> 4, want = (8 * q + j) * 64 + p;
            if (h[lane * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got T[%d][%d] want T[%d][%d]\n", lane, j, h[lane*4+j] / 64, h[lane*4+j] % 64, want / 64, want % 64); ++bad; }
        }
    printf("tr_read probe: %d mismatches\n", bad);
    return bad != 0;
}

// Split of f32 operands into bf16 parts for the bf16 matrix pipe (shared by k_pw_x3.hip and k_xdw_stream.hip).
#pragma once
#include "common.hpp"

namespace ams {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // native vector: stays in registers where HIP's uint4 struct may not

// 8 consecutive f32 -> bf16x8 parts (hi, mid, lo): successive bf16 roundings of the remainder.  Plain named vectors (an
// array of vectors filled element-wise lands in scratch memory).
__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& p0) {     // one part: plain bf16 rounding
    const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) p0[j] = (__bf16)f[j];
}
__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& p0, bf16x8& p1) {
    const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        p0[j] = h;
        p1[j] = (__bf16)(f[j] - (float)h);
    }
}
__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        const float r1 = f[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        p0[j] = h;
        p1[j] = m;
        p2[j] = (__bf16)(r1 - (float)m);
    }
}

}  // namespace ams

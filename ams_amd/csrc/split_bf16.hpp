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
// Three parts, written pair by pair so that every conversion is one v_cvt_pk_bf16_f32 on two values and every remainder one
// v_pk_add_f32 (9 instructions per pair; the element-wise form compiled to ~12 with single-value conversions and byte permutes).
// Same roundings: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), the subtractions are exact.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2s_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2s_t f = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
    const f32x2s_t hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const f32x2s_t r1 = f - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
    const f32x2s_t mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    const f32x2s_t r2 = r1 - mf;
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}
__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split_pair(u.x, u.y, h0, m0, l0);
    split_pair(u.z, u.w, h1, m1, l1);
    split_pair(v.x, v.y, h2, m2, l2);
    split_pair(v.z, v.w, h3, m3, l3);
    const u32x4 h = {h0, h1, h2, h3}, m = {m0, m1, m2, m3}, l = {l0, l1, l2, l3};
    p0 = __builtin_bit_cast(bf16x8, h);
    p1 = __builtin_bit_cast(bf16x8, m);
    p2 = __builtin_bit_cast(bf16x8, l);
}

}  // namespace ams

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

// ---- two fp16 parts, 22 significand bits: x ~ hi + lo * 2^-11 with hi = f16(x), lo = f16((x - hi) * 2^11) ----------------------------
// (AMS_MATMUL_SPLIT_F16).  hi carries 11 bits; the remainder x - hi is exact in f32 and at most half an ulp of hi, so scaled by 2^11 it
// is back in hi's binade or below — normal in fp16 whenever hi is — and lo keeps 11 more bits: |x - (hi + lo 2^-11)| <= 2^-22 |x| (worst
// case; 2^-23.8 rms).  The product of two such operands is formed as  hi*hi  +  2^-11 (hi*lo + lo*hi)  with the cross terms in an
// accumulator of their own (3 MFMAs, v_mfma_f32_16x16x32_f16; the dropped lo*lo term is <= 2^-22 of the product): per product ~3 2^-22
// worst case against 2^-24 for the three-part bf16 split (6 MFMAs) — below the f32 accumulation error of a K >= 64 contraction (measured:
// 512x1024 logits 4e-5 from f64 either way) — at 4 bytes per value instead of 6 where the parts are stored.  Values beyond fp16's range (|x| >= 65520) become inf: the engine uses the
// form only on layers whose operands are activations and weights of O(1).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr float kF16LoScale = 2048.f, kF16LoInv = 1.f / 2048.f;
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& h, unsigned& l) {
    const f32x2s_t f = {a, b};
    const f16x2_t hh = __builtin_convertvector(f, f16x2_t);                     // v_cvt_pk_f16_f32 (RNE)
    h = __builtin_bit_cast(unsigned, hh);
    const f32x2s_t s = f * (f32x2s_t){kF16LoScale, kF16LoScale};
    const f32x2s_t r = {fmaf((float)hh.x, -kF16LoScale, s.x), fmaf((float)hh.y, -kF16LoScale, s.y)};     // (x - hi) 2^11, exact
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
__device__ __forceinline__ void split8_f16(const float4& u, const float4& v, f16x8& p0, f16x8& p1) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair_f16(u.x, u.y, h0, l0);
    split_pair_f16(u.z, u.w, h1, l1);
    split_pair_f16(v.x, v.y, h2, l2);
    split_pair_f16(v.z, v.w, h3, l3);
    const u32x4 h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
    p0 = __builtin_bit_cast(f16x8, h);
    p1 = __builtin_bit_cast(f16x8, l);
}
// four values -> (hi, lo) as two dwords each
__device__ __forceinline__ void split4_f16(const float4& v, unsigned (&h)[2], unsigned (&l)[2]) {
    split_pair_f16(v.x, v.y, h[0], l[0]);
    split_pair_f16(v.z, v.w, h[1], l[1]);
}
// The same bits with the remainder taken by v_fma_mix_f32, which reads hi as fp16 straight from the packed dword: r = fma(f32(hi), -2^11, x 2^11)
// without the two v_cvt_f32_f16 per pair (VALU-bound kernels: the D-waves of k_xdw_wreg.hip)
__device__ __forceinline__ void split_pair_f16_mix(float a, float b, unsigned& h, unsigned& l) {
    const f32x2s_t f = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2_t));                      // v_cvt_pk_f16_f32 (RNE)
    const float sa = a * kF16LoScale, sb = b * kF16LoScale;
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(-kF16LoScale), "v"(sa));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(-kF16LoScale), "v"(sb));
    const f32x2s_t r = {ra, rb};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
__device__ __forceinline__ void split4_f16_mix(const float4& v, unsigned (&h)[2], unsigned (&l)[2]) {
    split_pair_f16_mix(v.x, v.y, h[0], l[0]);
    split_pair_f16_mix(v.z, v.w, h[1], l[1]);
}
// scalar form (weight panels): bit patterns of (hi, lo)
__device__ __forceinline__ void split1_f16(float v, unsigned short& h, unsigned short& l) {
    const _Float16 hh = (_Float16)v;
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, (_Float16)fmaf((float)hh, -kF16LoScale, v * kF16LoScale));
}
// acc (hi*hi) and accx (cross terms, scaled 2^11) -> the product
__device__ __forceinline__ f32x4 combine_f16(const f32x4& acc, const f32x4& accx) {
    return (f32x4){fmaf(accx[0], kF16LoInv, acc[0]), fmaf(accx[1], kF16LoInv, acc[1]), fmaf(accx[2], kF16LoInv, acc[2]), fmaf(accx[3], kF16LoInv, acc[3])};
}

}  // namespace ams

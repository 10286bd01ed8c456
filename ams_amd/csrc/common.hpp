// Shared device/host helpers for the AMS student kernels (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ams_hip.h"

namespace ams {

// ---- error plumbing ------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
// per-launch profiling hook: launchers name their main kernel (as rocprofv3 prints it, minus namespace/arguments)
void note_kernel(const char* name);

#define AMS_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::ams::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return AMS_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define AMS_CHECK_LAUNCH()                                                                    \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            ::ams::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return AMS_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define AMS_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            ::ams::set_error(__VA_ARGS__);                                                    \
            return AMS_E_INVALID;                                                             \
        }                                                                                     \
    } while (0)

#define RUN_RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// TF 'SAME' padding: out = ceil(in/stride), surplus pad goes after.
static inline void same_pad(int in, int k, int stride, int rate, int* out, int* before) {
    int o = (in + stride - 1) / stride;
    int eff = (k - 1) * rate + 1;
    int total = (o - 1) * stride + eff - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

// acc += v * w on 4 channels as two v_pk_fma_f32 (per element the same single rounding as fmaf): half the VALU issue slots of four
// v_fma_f32.  For VALU phases that are NOT interleaved instruction by instruction with MFMAs (tools/probes/README.md: packed f32
// between a wave's own MFMAs is slower than scalar).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma4_pk(float4& acc, const float4& v, const float4& w) {
    f32x2_t a0 = {acc.x, acc.y}, a1 = {acc.z, acc.w};
    const f32x2_t v0 = {v.x, v.y}, v1 = {v.z, v.w}, w0 = {w.x, w.y}, w1 = {w.z, w.w};
    a0 = __builtin_elementwise_fma(v0, w0, a0);
    a1 = __builtin_elementwise_fma(v1, w1, a1);
    acc = make_float4(a0.x, a0.y, a1.x, a1.y);
}

// v * sc + sh with the two roundings of an unfused multiply and add, as v_pk_mul_f32 + v_pk_add_f32 (two instructions for four values)
__device__ __forceinline__ float4 muladd4_pk(const float4& v, const float4& sc, const float4& sh) {
    const f32x2_t v0 = {v.x, v.y}, v1 = {v.z, v.w}, s0 = {sc.x, sc.y}, s1 = {sc.z, sc.w}, h0 = {sh.x, sh.y}, h1 = {sh.z, sh.w};
    const f32x2_t m0 = v0 * s0, m1 = v1 * s1;
    const f32x2_t r0 = m0 + h0, r1 = m1 + h1;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}

__device__ __forceinline__ float4 add4_pk(const float4& a, const float4& b) {
    const f32x2_t a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2_t r0 = a0 + b0, r1 = a1 + b1;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}
__device__ __forceinline__ float4 sub4_pk(const float4& a, const float4& b) {
    const f32x2_t a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2_t r0 = a0 - b0, r1 = a1 - b1;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}
__device__ __forceinline__ float4 mul4_pk(const float4& a, const float4& b) {
    const f32x2_t a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    const f32x2_t r0 = a0 * b0, r1 = a1 * b1;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}

constexpr int kNumXcd = 8;   // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only, never correctness)

// ---- device helpers --------------------------------------------------------------------------------
// branch-free: the activation is a clamp to [lo, hi] with wave-uniform bounds (scalar selects) — ONE v_med3_f32 per value (the
// fminf(fmaxf()) form is two instructions: hipcc cannot prove lo <= hi for run-time bounds; VALU instructions beside MFMAs are not
// free on this chip, tools/probes/README.md)
__device__ __forceinline__ float apply_act(float v, int act) {
    const float lo = act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f;
    const float hi = act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}

// XCD-aware block remap: consecutive *logical* ids land on the same XCD (same private L2), so blocks that
// share halos / operand panels hit in L2 instead of each XCD re-fetching them (guide T1, bijective form).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    unsigned q = nblocks / kNumXcd, r = nblocks % kNumXcd;
    unsigned xcd = bid % kNumXcd, slot = bid / kNumXcd;
    unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

}  // namespace ams

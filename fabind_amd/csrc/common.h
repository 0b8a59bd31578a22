// Shared device helpers for the fabind_amd HIP kernels (gfx950 / CDNA4 only: wave64, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FB_DT_F32 0
#define FB_DT_BF16 1

#define FB_ACT_NONE 0
#define FB_ACT_SILU 1
#define FB_ACT_RELU 2
#define FB_ACT_SIGMOID 3
#define FB_ACT_STORED_DERIV 4  // "derivative" operand already holds act'(pre): apply_dact(x) = x

typedef uint16_t bf16_t;  // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

extern "C" void fabind_set_error(const char* msg);

#define FB_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) {                            \
            fabind_set_error(hipGetErrorString(e__));       \
            return (int)e__;                                \
        }                                                   \
    } while (0)

#define FB_REQUIRE(cond, msg)          \
    do {                               \
        if (!(cond)) {                 \
            fabind_set_error(msg);     \
            return -1;                 \
        }                              \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// d/dx silu(x) = s(x) * (1 + x * (1 - s(x)))
__device__ __forceinline__ float dsilu_f(float x) {
    float s = sigmoid_f(x);
    return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case FB_ACT_SILU: return silu_f(x);
        case FB_ACT_RELU: return x > 0.f ? x : 0.f;
        case FB_ACT_SIGMOID: return sigmoid_f(x);
        default: return x;
    }
}
// derivative of act w.r.t. its pre-activation x
__device__ __forceinline__ float apply_dact(float x, int act) {
    switch (act) {
        case FB_ACT_SILU: return dsilu_f(x);
        case FB_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case FB_ACT_SIGMOID: { float s = sigmoid_f(x); return s * (1.f - s); }
        case FB_ACT_STORED_DERIV: return x;
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {  // all 64 lanes get the total
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// generic element load/store by runtime dtype
__device__ __forceinline__ float ld_any(const void* p, int dt, size_t i) {
    return dt == FB_DT_F32 ? ((const float*)p)[i] : bf16_to_f32(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st_any(void* p, int dt, size_t i, float v) {
    if (dt == FB_DT_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f32_to_bf16(v);
}

// 4 consecutive elements (16-byte aligned for f32, 8-byte for bf16)
__device__ __forceinline__ float4 ld4_any(const void* p, int dt, size_t i) {
    if (dt == FB_DT_F32) return *(const float4*)((const float*)p + i);
    ushort4 u = *(const ushort4*)((const bf16_t*)p + i);
    return make_float4(bf16_to_f32(u.x), bf16_to_f32(u.y), bf16_to_f32(u.z), bf16_to_f32(u.w));
}
__device__ __forceinline__ void st4_any(void* p, int dt, size_t i, float4 v) {
    if (dt == FB_DT_F32) { *(float4*)((float*)p + i) = v; return; }
    ushort4 u; u.x = f32_to_bf16(v.x); u.y = f32_to_bf16(v.y); u.z = f32_to_bf16(v.z); u.w = f32_to_bf16(v.w);
    *(ushort4*)((bf16_t*)p + i) = u;
}

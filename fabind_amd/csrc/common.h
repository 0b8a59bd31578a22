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

// Function attribute: no packed fp32 math (v_pk_*_f32) in this kernel (the LAS step: attn.hip).  A subtarget feature of the DEVICE pass; the host
// pass of the same source does not know it.
#if defined(__HIP_DEVICE_COMPILE__)
#define FB_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define FB_NO_PACKED_F32
#endif

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

// round-to-nearest-even through gfx950's v_cvt_pk_bf16_f32 (two floats -> one packed dword)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *(const uint32_t*)&b;
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// v_exp_f32 + v_rcp_f32 (1 ulp each); an IEEE division here made the gather / epilogue kernels VALU-bound
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
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

// counter-based 32-bit hash (dropout masks evaluated in kernels: fused edge kernels, GEMM epilogue)
__device__ __forceinline__ uint32_t fb_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
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
    *(uint2*)((bf16_t*)p + i) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
}
// 8 consecutive elements per lane (16-byte bf16 / 2 x 16-byte fp32 accesses)
struct F8 { float v[8]; };
__device__ __forceinline__ F8 ld8_any(const void* p, int dt, size_t i) {
    F8 r;
    if (dt == FB_DT_F32) {
        const float4 a = *(const float4*)((const float*)p + i), b = *(const float4*)((const float*)p + i + 4);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    } else {
        const uint4 u = *(const uint4*)((const bf16_t*)p + i);
        r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
        r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
        r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
        r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    return r;
}
__device__ __forceinline__ void st8_any(void* p, int dt, size_t i, const F8& v) {
    if (dt == FB_DT_F32) {
        *(float4*)((float*)p + i) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
        *(float4*)((float*)p + i + 4) = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]);
        return;
    }
    uint4 u;
    u.x = pack2_bf16(v.v[0], v.v[1]);
    u.y = pack2_bf16(v.v[2], v.v[3]);
    u.z = pack2_bf16(v.v[4], v.v[5]);
    u.w = pack2_bf16(v.v[6], v.v[7]);
    *(uint4*)((bf16_t*)p + i) = u;
}

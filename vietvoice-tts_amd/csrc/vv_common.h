// Shared device helpers for the gfx950 kernels (wave64, MFMA fragment types, LDS swizzles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define VV_WAVE 64

// dtype codes used across the C ABI
enum { VV_F32 = 0, VV_BF16 = 1 };
// activation codes
enum { VV_ACT_NONE = 0, VV_ACT_GELU_TANH = 1, VV_ACT_GELU_ERF = 2, VV_ACT_SILU = 3, VV_ACT_MISH = 4 };

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  The LDS destination is
// wave-uniform base + lane*16; the per-lane part is the SOURCE address.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Swizzle for tiles with 128-byte rows (8 chunks of 16 B): chunk' = c ^ ((row>>1)&7).
// Conflict-free for ds_read_b128 both when a lane group reads 16 distinct rows (mod 16) of one
// chunk (32x32 operands) and when it reads rows 0-15 of two adjacent chunks (16x16x32 operands).
__device__ __forceinline__ int swz128(int row, int chunk) { return (row << 7) | ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case VV_ACT_GELU_TANH: {
            float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.0f + tanhf(u));
        }
        case VV_ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
        case VV_ACT_SILU: return x / (1.0f + expf(-x));
        case VV_ACT_MISH: {
            // x * tanh(softplus(x)) = x * t / (t + 2),  t = e^x (e^x + 2)   (one v_exp + one v_rcp)
            if (x > 20.0f) return x;
            const float n = __builtin_amdgcn_exp2f(1.4426950408889634f * x);
            const float t = n * (n + 2.0f);
            return x * t * __builtin_amdgcn_rcpf(t + 2.0f);
        }
        default: return x;
    }
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, float a, float b, float c, float d) {
    bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
    *(bf16x4*)p = v;
}
template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 load4<bf16>(const bf16* p) {
    bf16x4 v = *(const bf16x4*)p;
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// non-temporal forms for operands that are streamed exactly once
template <typename T> __device__ __forceinline__ void store4_nt(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4_nt<float>(float* p, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    const f4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, (f4*)p);
}
template <> __device__ __forceinline__ void store4_nt<bf16>(bf16* p, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    const bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
    __builtin_nontemporal_store(*(const u2*)&v, (u2*)p);
}
template <typename T> __device__ __forceinline__ float4 load4_nt(const T* p);
template <> __device__ __forceinline__ float4 load4_nt<float>(const float* p) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    const f4 v = __builtin_nontemporal_load((const f4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ float4 load4_nt<bf16>(const bf16* p) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    const u2 w = __builtin_nontemporal_load((const u2*)p);
    return make_float4(__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u));
}

// Shared device helpers for the gfx950 kernels of libfw_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fw_mi355x.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define FW_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define FW_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-byte async global->LDS copy: LDS destination = wave-uniform base + lane*16 (gfx950 global_load_lds_dwordx4)
#define FW_GLDS16(gptr, ldsbase) __builtin_amdgcn_global_load_lds(FW_GLB_PTR(gptr), FW_LDS_PTR(ldsbase), 16, 0, 0)

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even fp32 -> bf16 bits (NaN preserved as quiet NaN)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);    // hardware convert on gfx950
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// two fp32 -> packed bf16x2 with the gfx950 hardware convert (v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float fw_gelu_tanh(float x) {
    // 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715x^3)))  == x * sigmoid(2u)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}
__device__ __forceinline__ float fw_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float fw_silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float fw_apply_act(float v, int act) {
    switch (act) {
        case FW_ACT_RELU: return v > 0.f ? v : 0.f;
        case FW_ACT_GELU_TANH: return fw_gelu_tanh(v);
        case FW_ACT_GELU_ERF: return fw_gelu_erf(v);
        case FW_ACT_SILU: return fw_silu(v);
        default: return v;
    }
}

// Per-column affine of the GEMM epilogues (gate / LayerScale / modulation): ONE fused multiply-add, spelled out so that every GEMM
// kernel rounds it the same way -- rows that fall into the 128x128 tail kernel in one call and into a 256x256 tile in another
// (the merged CFG pass stacks two samples along the rows) must come out bit-identical.  Left to the compiler's contraction the
// kernels disagreed in the last bit.
__device__ __forceinline__ float fw_affine(float v, float g1, float g0) { return __builtin_fmaf(v, g1, g0); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// host-side error plumbing (api.cpp)
void fw_set_error(const char* msg);
int fw_get_option(int opt);

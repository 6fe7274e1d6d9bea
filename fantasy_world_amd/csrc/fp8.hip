// fp8 linear: the reference's only fp8 definition, AutoWrappedLinear.fp8_linear
// (FantasyWorld/diffsynth_wan22/vram_management/layers.py:115-151):
//     scale_a[m] = max(max_k |x[m][k]| / 448, 1)                 per-row activation scale, never below 1
//     xq = e4m3(x / (scale_a + 1e-8)),  wq = e4m3(w)              OCP e4m3fn (gfx950's native fp8), weights cast raw
//     y  = (xq wq^T) * scale_a + bf16(bias)   -> out dtype        torch._scaled_mm(xq, wq^T, scale_a, 1, bias, out_dtype)
// This file: the row quantiser (also used with scale = 1 to cast the weights once at pack time).  The GEMM lives in gemm_fp8.hip.
#include "fw_common.h"

namespace {

// The reference divides in fp32 and casts with round-to-nearest-even; a reciprocal-multiply division would flip values that sit on
// a rounding boundary of the 3-bit mantissa, so this file is compiled WITHOUT -ffast-math (csrc/build.sh): `/` is IEEE here.
__device__ __forceinline__ float fp8_prescale(float v, float denom) { return v / denom; }

// xq[m][k] = e4m3(x[m][k] / (scale[m] + 1e-8)), scale[m] = fixed_scale > 0 ? fixed_scale : max(amax_m / 448, 1).  One wave per row.
// amax_in != nullptr: the row maximum is SUPPLIED (the full row's, all-reduced over the tensor-parallel ranks that each hold a K-slice)
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                             int64_t ldq, float* __restrict__ scale, int M, int K, int raw,
                                                             const float* __restrict__ amax_in) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint16_t* src = x + (int64_t)row * ldx;
    float denom = 1.0f;
    if (!raw) {
        float amax = 0.f;
        if (amax_in) {
            amax = amax_in[row];
        } else {
            for (int k = lane * 8; k < K; k += 512) {
                const u32x4_t v = *(const u32x4_t*)(src + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    amax = fmaxf(amax, fabsf(__uint_as_float(v[i] << 16)));
                    amax = fmaxf(amax, fabsf(__uint_as_float(v[i] & 0xffff0000u)));
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        }
        // x_max is taken in the input dtype (bf16: exact), / 448 in that dtype as well (layers.py:126-133), then .float()
        const float s = fmaxf(bf16_bits_to_f32(f32_to_bf16_bits(fp8_prescale(amax, 448.0f))), 1.0f);
        if (lane == 0) scale[row] = s;
        denom = s + 1e-8f;
    }
    uint8_t* dst = q + (int64_t)row * ldq;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4_t v = *(const u32x4_t*)(src + k);
        float f[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
        if (!raw) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = fp8_prescale(f[i], denom);
        }
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        u32x2_t w = {(uint32_t)lo, (uint32_t)hi};
        *(u32x2_t*)(dst + k) = w;
    }
}

// out[m] = max_k |x[m][k]| of a bf16 slice (exact in fp32): one wave per row
__global__ __launch_bounds__(256) void row_absmax_kernel(const uint16_t* __restrict__ x, int64_t ldx, int rows, int width,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* src = x + (int64_t)row * ldx;
    float amax = 0.f;
    for (int k = lane * 8; k < width; k += 512) {
        const u32x4_t v = *(const u32x4_t*)(src + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            amax = fmaxf(amax, fabsf(__uint_as_float(v[i] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(v[i] & 0xffff0000u)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) out[row] = amax;
}

}  // namespace

extern "C" int fw_fp8_quant_rows(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int M, int K, int raw,
                                 void* stream) {
    if (!x || !q || (!raw && !scale) || K <= 0 || (K % 8) || (ldx % 8) || (ldq % 8) || (((uintptr_t)x) & 15) || (((uintptr_t)q) & 7)) {
        fw_set_error("fw_fp8_quant_rows: K, ldx, ldq must be multiples of 8, x 16-byte and q 8-byte aligned"); return FW_E_BADARG; }
    if (M <= 0) return 0;
    hipLaunchKernelGGL(fp8_quant_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, q, ldq, scale, M, K, raw,
                       (const float*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int fw_fp8_quant_rows_amax(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, const float* amax, float* scale, int M,
                                      int K, void* stream) {
    if (!x || !q || !amax || !scale || K <= 0 || (K % 8) || (ldx % 8) || (ldq % 8) || (((uintptr_t)x) & 15) || (((uintptr_t)q) & 7)) {
        fw_set_error("fw_fp8_quant_rows_amax: amax / scale required; K, ldx, ldq multiples of 8, x 16-byte and q 8-byte aligned"); return FW_E_BADARG; }
    if (M <= 0) return 0;
    hipLaunchKernelGGL(fp8_quant_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, q, ldq, scale, M, K, 0,
                       amax);
    return (int)hipGetLastError();
}

extern "C" int fw_row_absmax(const uint16_t* x, int64_t ldx, int rows, int width, float* out, void* stream) {
    if (rows <= 0) return 0;
    if (!x || !out || (width % 8) || (ldx % 8) || (((uintptr_t)x) & 15)) { fw_set_error("fw_row_absmax: width % 8 == 0 and 16-B alignment required"); return FW_E_BADARG; }
    hipLaunchKernelGGL(row_absmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, width, out);
    return (int)hipGetLastError();
}


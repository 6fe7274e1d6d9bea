// What v_cvt_pk_u8_f32 does on gfx950 (rounding, saturation, NaN) and what it costs beside v_exp_f32 + v_cvt_pk_fp8_f32: the facts the
// "e4m3 bits of 2^x are 8(x+7), piecewise linear" softmax of attention_fp8.hip rests on.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/cvt_probe tools/probes/cvt_pk_u8_probe.hip && /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>

__global__ void semantics(const float* in, uint32_t* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}

template <int MODE>
__global__ __launch_bounds__(256) void cost(float* io, uint32_t* sink, int iters, long long* cycles) {
    float s[32];
    for (int r = 0; r < 32; ++r) s[r] = io[threadIdx.x * 32 + r];
    uint32_t acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint32_t w[8];
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float a0 = __builtin_amdgcn_exp2f(s[4 * g]), a1 = __builtin_amdgcn_exp2f(s[4 * g + 1]);
                const float a2 = __builtin_amdgcn_exp2f(s[4 * g + 2]), a3 = __builtin_amdgcn_exp2f(s[4 * g + 3]);
                int x = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
                w[g] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, x, true);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(s[4 * g], 0, 0);
                x = __builtin_amdgcn_cvt_pk_u8_f32(s[4 * g + 1], 1, x);
                x = __builtin_amdgcn_cvt_pk_u8_f32(s[4 * g + 2], 2, x);
                w[g] = __builtin_amdgcn_cvt_pk_u8_f32(s[4 * g + 3], 3, x);
            }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) acc ^= w[g];
#pragma unroll
        for (int r = 0; r < 32; ++r) asm volatile("" : "+v"(s[r]));
    }
    const long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[MODE] = t1 - t0;
}

int main() {
    const float vals[] = {-1.0e30f, -3.7f, -0.6f, -0.4f, 0.0f, 0.3f, 0.5f, 0.7f, 1.5f, 2.5f, 3.5f, 3.49f, 119.5f, 120.5f, 126.4f, 126.5f, 127.5f,
                          254.4f, 254.5f, 255.0f, 255.5f, 256.0f, 300.0f, 1.0e30f, NAN, INFINITY, -INFINITY, 7.999f, 8.0f, 8.001f, 55.5f, 56.5f};
    const int n = sizeof(vals) / sizeof(float);
    float* din; uint32_t* dout;
    hipMalloc(&din, sizeof(vals)); hipMalloc(&dout, n * 4);
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, din, dout, n);
    uint32_t out[64];
    hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32(x, byte 1, 0xAABBCCDD):\n");
    for (int i = 0; i < n; ++i) printf("  x = %-12g -> word %08x  byte %3u\n", vals[i], out[i], (out[i] >> 8) & 0xff);
    // cost: one wave per SIMD (256 threads, one work-group per CU), 32 scores -> 8 words, 4096 iterations
    float* io; uint32_t* sink; long long* cyc;
    hipMalloc(&io, 256 * 256 * 32 * 4); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 16);
    hipMemset(io, 0, 256 * 256 * 32 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(cost<0>, dim3(256), dim3(256), 0, 0, io, sink, 4096, cyc);
        hipLaunchKernelGGL(cost<1>, dim3(256), dim3(256), 0, 0, io, sink, 4096, cyc);
        hipDeviceSynchronize();
    }
    long long c[2];
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("32 scores -> 8 e4m3 words, one wave per SIMD, s_memtime ticks per iteration (100 MHz constant clock: relative only):\n");
    printf("  32 v_exp_f32 + 16 v_cvt_pk_fp8_f32 : %.2f\n  32 v_cvt_pk_u8_f32                 : %.2f\n", c[0] / 4096.0, c[1] / 4096.0);
    return 0;
}

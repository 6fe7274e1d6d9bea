// Probe: issue cost (shader cycles per wave-instruction, one wave per SIMD) of the VALU ops of the attention softmax, alone and
// in the shadow of v_mfma_f32_32x32x16_bf16.  Build: hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define REP16(X) X X X X X X X X X X X X X X X X

// MODE: 0 v_exp_f32, 1 v_fma_f32, 2 v_add_f32, 3 v_cvt_pk_bf16_f32, 4 v_pk_add_f32, 5 v_max3_f32, 6 v_pk_fma_f32,
//       7 v_dot2_f32_bf16, 8 v_exp_f16, 10 MFMA alone, 11 MFMA + 2 exp per gap, 12 MFMA + 4 fma per gap, 13 MFMA + 1 exp + 3 valu
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
    float a[16], b = 1.0001f + threadIdx.x * 1e-7f, c = 0.5f;
    for (int i = 0; i < 16; ++i) a[i] = 0.001f * (i + 1) + threadIdx.x * 1e-6f;
    f32x16_t acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8_t fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));) }
        if (MODE == 0) { }
        if (MODE == 1) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
        if (MODE == 2) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
        if (MODE == 3) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
        if (MODE == 4) { _Pragma("unroll") for (int i = 0; i < 16; i += 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i]) : "v"(*(double*)&a[(i + 2) & 15])); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i]) : "v"(*(double*)&a[(i + 4) & 15])); } }
        if (MODE == 5) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
        if (MODE == 6) { _Pragma("unroll") for (int i = 0; i < 16; i += 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i]) : "v"(*(double*)&a[(i + 2) & 15])); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i]) : "v"(*(double*)&a[(i + 4) & 15])); } }
        if (MODE == 7) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c)); }
        if (MODE == 8) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i])); }
        if (MODE == 9) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); }
        if (MODE >= 10) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                if (MODE == 11) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * m])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * m + 1])); }
                if (MODE == 12) { for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(4 * m + j) & 15]) : "v"(b), "v"(c)); }
                if (MODE == 13) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[m])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[m + 8]) : "v"(c)); asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[(m + 4) & 15]) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(m + 9) & 15]) : "v"(c)); }
                if (MODE == 14) { for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(8 * m + j) & 15]) : "v"(b), "v"(c)); }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc1, 0, 0, 0);
                if (MODE == 11) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[(2 * m + 5) & 15])); asm volatile("v_exp_f32 %0, %0" : "+v"(a[(2 * m + 9) & 15])); }
                if (MODE == 12) { for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(4 * m + j + 7) & 15]) : "v"(b), "v"(c)); }
                if (MODE == 13) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[(m + 3) & 15])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(m + 11) & 15]) : "v"(c)); asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[(m + 6) & 15]) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(m + 13) & 15]) : "v"(c)); }
                if (MODE == 14) { for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(8 * m + j + 5) & 15]) : "v"(b), "v"(c)); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 100);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.2f ticks per instruction-group (%d per iter)\n", name, (double)c / iters / per_iter, per_iter);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    run<0>("v_exp_f32 dependent chain (latency)", 16, out, cyc);
    run<9>("v_exp_f32 x16 independent", 16, out, cyc);
    run<8>("v_exp_f16 x16 independent", 16, out, cyc);
    run<1>("v_fma_f32 x16 independent", 16, out, cyc);
    run<2>("v_add_f32 x16 independent", 16, out, cyc);
    run<3>("v_cvt_pk_bf16_f32 x16", 16, out, cyc);
    run<4>("v_pk_add_f32 x16", 16, out, cyc);
    run<6>("v_pk_fma_f32 x16", 16, out, cyc);
    run<5>("v_max3_f32 x16", 16, out, cyc);
    run<7>("v_dot2_f32_bf16 x16", 16, out, cyc);
    run<10>("MFMA 32x32x16 bf16 alone (per MFMA)", 16, out, cyc);
    run<11>("MFMA + 2 v_exp per gap (per MFMA)", 16, out, cyc);
    run<12>("MFMA + 4 v_fma per gap (per MFMA)", 16, out, cyc);
    run<14>("MFMA + 8 v_fma per gap (per MFMA)", 16, out, cyc);
    run<13>("MFMA + exp,add,cvt,add per gap (per MFMA)", 16, out, cyc);
    return 0;
}

// Shared pieces of the bf16 GEMM translation units (gemm.hip: 128x128 kernel, four-slot ping-pong kernel incl. the implicit-GEMM
// convolution, four-wave kernel, launchers; gemm_pp.hip: the two-slot ping-pong kernels of round 4): argument block, tile constants,
// the fused epilogue of the 256x256 tiles.  Everything here is header-only and has internal linkage.
#pragma once
#include "fw_common.h"
#include <stdlib.h>
#include <type_traits>

namespace fwgemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int GROUP_M = 8;

// Implicit-GEMM convolution (fw_conv_gemm_bf16): the A operand is never materialised.  A is the channels-last feature map
// x [T*H*W][C] (lda = its row stride), GEMM row r = output pixel (t, yo, xo), GEMM column k = tap * C + c with tap = (dt*kh + dy)*kw
// + dx.  C is a multiple of the k-slab (64), so a slab lies inside ONE tap: the LDS-DMA source of a tile row is just another row of x
// (or 128 B of zeros outside the volume) -- the gather costs an address computation per DMA piece and no memory traffic.
struct ConvGeom {
    int T, H, W, Ho, Wo;                 // stored map, output map
    int kt, kh, kw, sh, sw, ph, pw;      // taps, spatial stride / padding (time: causal, kt - 1 frames of history)
    int ups;                             // log2 of the nearest-neighbour up-sampling the convolution sees (0 or 1)
    int t0;                              // first output frame
    int cpt;                             // k-slabs per tap = C / 64
};

struct GemmArgs {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;
    void* C; int64_t ldc; int out_dtype;
    int M, N, K;
    const float* bias; int act; const float* g1; const float* g0;
    const void* res; int64_t ldr; int res_dtype;
    int tiles_m, tiles_n;
    int group_m;                         // M-tiles per group of the tile order of the ping-pong kernel
    ConvGeom cv;                         // read by the CONV instantiations only
};


// fw_apply_act with the activation as a compile-time constant (straight-line code inside the unrolled element loops)
template <int ACT> __device__ __forceinline__ float fw_apply_act_ct(float v) {
    if (ACT == FW_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == FW_ACT_GELU_TANH) return fw_gelu_tanh(v);
    if (ACT == FW_ACT_GELU_ERF) return fw_gelu_erf(v);
    if (ACT == FW_ACT_SILU) return fw_silu(v);
    return v;
}


// ---- fused epilogue of the 256x256 kernels, through LDS: each wave transposes its 128x64 result in two 64-row passes
// through a private 16 KiB region (fp32, row stride 256 B), so that global traffic is row-contiguous 16-B (fp32) / 8-B
// (bf16) per lane: residual loads and output stores touch whole 128-B lines instead of 2-4 B per lane at a row stride.
// bias / activation / per-column affine are applied on the way in (column == lane in the accumulator layout).
//
// Activation, residual type and output type are COMPILE-TIME parameters of the body, selected once per wave (round 2, read off
// the ISA: with `switch (p.act)` inside the element loop every one of the 256 accumulator values of a wave walked a chain of scalar
// compares and taken branches -- 35 000 cycles = 17-19 us of a 145 us tile, measured with tools/gemm_timeline.py and as the
// intercept of time against K, tools/probes/gemm_intercept.py: 20 us fixed per tile plain, 32 us with the fp32 residual).

template <int ACT, int RES, int OUT>
__device__ __forceinline__ void epilogue_256_body(const GemmArgs& p, char* reg, f32x16_t (&acc)[4][2], int grp, int wn,
                                                  int fi, int hi, int lane, int m0, int n0) {
    float bias2[2], g12[2], g02[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = n0 + wn * 64 + nb * 32 + fi;
        const bool ok = col < p.N;
        bias2[nb] = (p.bias && ok) ? p.bias[col] : 0.f;
        g12[nb] = (p.g1 && ok) ? p.g1[col] : 1.f;
        g02[nb] = (p.g0 && ok) ? p.g0[col] : 0.f;
    }
    const int rl = lane >> 4;              // row inside a 4-row read group
    const int c4 = (lane & 15) * 4;        // first of this lane's 4 columns
    const int gcol = n0 + wn * 64 + c4;
    const bool col_ok = gcol < p.N;
    // Residual values of a 64-row pass: ALL 16 loads are in flight before the LDS transpose (16 KiB per wave, 128 KiB per CU), and
    // the loads of pass 1 are issued from inside the store loop of pass 0 -- each as soon as the register that held pass 0's value
    // is free and BEFORE the store of the same rows: their latency runs under pass 0's stores and pass 1's transpose, and waiting
    // for them never has to wait for a later store (vmcnt retires in order).
    f32x4_t rv[16];
    u32x2_t rw[16];
    auto load_res = [&](int it, int row) __attribute__((always_inline)) {
        const bool ok = row < p.M && col_ok;
        if (RES == FW_DT_F32) {
            rv[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ok) rv[it] = *(const f32x4_t*)((const float*)p.res + (int64_t)row * p.ldr + gcol);
        } else if (RES == FW_DT_BF16) {
            rw[it] = u32x2_t{0u, 0u};
            if (ok) rw[it] = *(const u32x2_t*)((const uint16_t*)p.res + (int64_t)row * p.ldr + gcol);
        }
    };
    if (RES != FW_DT_NONE) {
#pragma unroll
        for (int it = 0; it < 16; ++it) load_res(it, m0 + grp * 128 + it * 4 + rl);
    }
    auto pass = [&](auto q_tag) __attribute__((always_inline)) {
        constexpr int q = decltype(q_tag)::value;              // compile-time: acc[] must never be indexed dynamically
#pragma unroll
        for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[2 * q + rb2][nb][r] + bias2[nb];
                    v = fw_apply_act_ct<ACT>(v);
                    v = fw_affine(v, g12[nb], g02[nb]);
                    const int row_l = rb2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    *(float*)(reg + row_l * 256 + (nb * 32 + fi) * 4) = v;
                }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row_l = it * 4 + rl;
            f32x4_t v = *(const f32x4_t*)(reg + row_l * 256 + c4 * 4);
            const int row = m0 + grp * 128 + q * 64 + row_l;
            if (RES == FW_DT_F32) {
                v += rv[it];
            } else if (RES == FW_DT_BF16) {
                v[0] += __uint_as_float(rw[it][0] << 16); v[1] += __uint_as_float(rw[it][0] & 0xffff0000u);
                v[2] += __uint_as_float(rw[it][1] << 16); v[3] += __uint_as_float(rw[it][1] & 0xffff0000u);
            }
            if (q == 0 && RES != FW_DT_NONE) load_res(it, row + 64);
            if (row < p.M && col_ok) {
                if (OUT == FW_DT_F32) {
                    *(f32x4_t*)((float*)p.C + (int64_t)row * p.ldc + gcol) = v;
                } else {
                    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *(u32x2_t*)((uint16_t*)p.C + (int64_t)row * p.ldc + gcol) = o;
                }
            }
        }
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
}

// Dispatch (wave-uniform, once per wave): 5 activations x 3 residual types x 2 output types = 30 straight-line bodies.
template <int RES, int OUT>
__device__ __forceinline__ void epilogue_256_any_act(const GemmArgs& p, char* reg, f32x16_t (&acc)[4][2], int grp, int wn,
                                                     int fi, int hi, int lane, int m0, int n0) {
    switch (p.act) {
        case FW_ACT_RELU: epilogue_256_body<FW_ACT_RELU, RES, OUT>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0); break;
        case FW_ACT_GELU_TANH: epilogue_256_body<FW_ACT_GELU_TANH, RES, OUT>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0); break;
        case FW_ACT_GELU_ERF: epilogue_256_body<FW_ACT_GELU_ERF, RES, OUT>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0); break;
        case FW_ACT_SILU: epilogue_256_body<FW_ACT_SILU, RES, OUT>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0); break;
        default: epilogue_256_body<FW_ACT_NONE, RES, OUT>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0); break;
    }
}

__device__ __forceinline__ void epilogue_256(const GemmArgs& p, char* smem, f32x16_t (&acc)[4][2], int wave, int grp, int wn,
                                             int fi, int hi, int lane, int m0, int n0) {
    char* reg = smem + wave * 16384;
    const bool f32out = p.out_dtype == FW_DT_F32;
    if (p.res_dtype == FW_DT_NONE) {
        if (f32out) epilogue_256_any_act<FW_DT_NONE, FW_DT_F32>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0);
        else epilogue_256_any_act<FW_DT_NONE, FW_DT_BF16>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0);
    } else if (p.res_dtype == FW_DT_F32) {
        if (f32out) epilogue_256_any_act<FW_DT_F32, FW_DT_F32>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0);
        else epilogue_256_any_act<FW_DT_F32, FW_DT_BF16>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0);
    } else {
        if (f32out) epilogue_256_any_act<FW_DT_BF16, FW_DT_F32>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0);
        else epilogue_256_any_act<FW_DT_BF16, FW_DT_BF16>(p, reg, acc, grp, wn, fi, hi, lane, m0, n0);
    }
}

constexpr int TM = 256, TN = 256;
constexpr int STAGE2 = (TM + TN) * BK * 2;   // 64 KiB

// Barriers are raw s_barrier (inline asm): a __syncthreads() would drain the DMA queue (vmcnt(0)) at every barrier.
#define FW_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N> __device__ __forceinline__ void fw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace fwgemm

// gemm_pp.hip: launch the two-slot ping-pong kernel variant `kern` (6 / 7 / 8 / 9 ...) on a prepared argument block; returns false when
// the variant does not exist or the operands do not qualify (the caller then falls back to the four-slot kernel)
bool fw_launch_gemm_pp(const fwgemm::GemmArgs& p, int kern, int var, hipStream_t st);

// bf16 MFMA GEMM with fused epilogue for gfx950:  C = epi(A[M,K] * W[N,K]^T)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as 2x2 v_mfma_f32_32x32x16_bf16
// accumulators.  A and W k-slabs are streamed HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round
// trip), two LDS stages, one barrier per k-slab.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is
// XOR-swizzled with ((row>>1)&7) so that the ds_read_b128 fragment reads of a 16-lane service group hit 16
// distinct 16-B slots of the 256-B bank row (conflict free).  Because global_load_lds writes lane-linear,
// the swizzle is applied to the per-lane SOURCE address and again on the read (both sides or neither).
// Work-group ids are remapped so each XCD (private L2) owns a contiguous range of tiles, grouped 8 M-tiles
// deep so concurrently resident tiles share A bands and W panels in L2.
#include "fw_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int GROUP_M = 8;

struct GemmArgs {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;
    void* C; int64_t ldc; int out_dtype;
    int M, N, K;
    const float* bias; int act; const float* g1; const float* g0;
    const void* res; int64_t ldr; int res_dtype;
    int tiles_m, tiles_n;
};

__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- work-group -> tile mapping: XCD-contiguous, then GROUP_M-deep grouped order -------------------
    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;   // bijective for any nwg
    }
    int tm, tn;
    {
        const int per_group = GROUP_M * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- global -> LDS staging addresses ---------------------------------------------------------------
    // Each wave issues 4 pieces (1 KiB = 8 rows x 128 B) for A and 4 for W per stage.  Piece pc covers tile rows
    // 8*pc .. 8*pc+7; lane -> (row = 8*pc + lane/8, physical chunk = lane%8); logical chunk = phys ^ ((row>>1)&7).
    const uint16_t* ag[4];
    const uint16_t* wg_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave * 4 + i;
        const int row = pc * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int ar = min(m0 + row, p.M - 1);
        const int wr = min(n0 + row, p.N - 1);
        ag[i] = p.A + (int64_t)ar * p.lda + chunk * 8;
        wg_[i] = p.W + (int64_t)wr * p.ldw + chunk * 8;
    }

    auto stage = [&](int s) {
        char* a_lds = smem + s * STAGE_BYTES;
        char* b_lds = a_lds + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave * 4 + i;
            FW_GLDS16(ag[i], a_lds + pc * 1024);
            FW_GLDS16(wg_[i], b_lds + pc * 1024);
            ag[i] += BK;
            wg_[i] += BK;
        }
    };

    // ---- fragment read offsets -------------------------------------------------------------------------
    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;
    const int a_row_off = (wm * 64 + fi) * 128;                 // + rb*32*128
    const int b_row_off = BM * BK * 2 + (wn * 64 + fi) * 128;   // + nb*32*128

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage((kt + 1) & 1);
        const char* base = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t a0 = *(const bf16x8_t*)(base + a_row_off + coff[ks]);
            bf16x8_t a1 = *(const bf16x8_t*)(base + a_row_off + 32 * 128 + coff[ks]);
            bf16x8_t b0 = *(const bf16x8_t*)(base + b_row_off + coff[ks]);
            bf16x8_t b1 = *(const bf16x8_t*)(base + b_row_off + 32 * 128 + coff[ks]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();   // next slab landed (vmcnt(0)) and every wave is done reading this one
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = n0 + wn * 64 + nb * 32 + fi;
        const bool col_ok = col < p.N;
        const float bias = (p.bias && col_ok) ? p.bias[col] : 0.f;
        const float g1 = (p.g1 && col_ok) ? p.g1[col] : 1.f;
        const float g0 = (p.g0 && col_ok) ? p.g0[col] : 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < p.M && col_ok) {
                    float v = acc[rb][nb][r] + bias;
                    v = fw_apply_act(v, p.act);
                    v = v * g1 + g0;
                    if (p.res_dtype == FW_DT_F32) v += ((const float*)p.res)[(int64_t)row * p.ldr + col];
                    else if (p.res_dtype == FW_DT_BF16) v += bf16_bits_to_f32(((const uint16_t*)p.res)[(int64_t)row * p.ldr + col]);
                    if (p.out_dtype == FW_DT_F32) ((float*)p.C)[(int64_t)row * p.ldc + col] = v;
                    else ((uint16_t*)p.C)[(int64_t)row * p.ldc + col] = f32_to_bf16_bits(v);
                }
            }
        }
    }
}


// ---- fused epilogue of the 256x256 kernels, through LDS: each wave transposes its 128x64 result in two 64-row passes
// through a private 16 KiB region (fp32, row stride 256 B), so that global traffic is row-contiguous 16-B (fp32) / 8-B
// (bf16) per lane: residual loads and output stores touch whole 128-B lines instead of 2-4 B per lane at a row stride.
// bias / activation / per-column affine are applied on the way in (column == lane in the accumulator layout).
__device__ __forceinline__ void epilogue_256(const GemmArgs& p, char* smem, f32x16_t (&acc)[4][2], int wave, int grp, int wn,
                                             int fi, int hi, int lane, int m0, int n0) {
    char* reg = smem + wave * 16384;
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
    // One 64-row pass: (1) ALL 16 residual loads of the pass are issued first -- 16 KiB per wave, 128 KiB per CU in flight, so the
    // HBM latency of the fp32 stream is paid once per pass and runs under the LDS transpose instead of once per 4 rows --
    // (2) accumulators -> LDS with bias / activation / affine, (3) row-contiguous read-back, + residual, 16-B stores.
    auto pass = [&](auto q_tag, auto res_tag) {
        constexpr int q = decltype(q_tag)::value;              // compile-time: acc[] must never be indexed dynamically
        constexpr int RES = decltype(res_tag)::value;          // FW_DT_NONE / FW_DT_BF16 / FW_DT_F32
        f32x4_t rv[16];
        u32x2_t rw[16];
        if (RES != FW_DT_NONE) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = m0 + grp * 128 + q * 64 + it * 4 + rl;
                const bool ok = row < p.M && col_ok;
                if (RES == FW_DT_F32) {
                    rv[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (ok) rv[it] = *(const f32x4_t*)((const float*)p.res + (int64_t)row * p.ldr + gcol);
                } else {
                    rw[it] = u32x2_t{0u, 0u};
                    if (ok) rw[it] = *(const u32x2_t*)((const uint16_t*)p.res + (int64_t)row * p.ldr + gcol);
                }
            }
        }
#pragma unroll
        for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[2 * q + rb2][nb][r] + bias2[nb];
                    v = fw_apply_act(v, p.act);
                    v = v * g12[nb] + g02[nb];
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
            if (row < p.M && col_ok) {
                if (p.out_dtype == FW_DT_F32) {
                    *(f32x4_t*)((float*)p.C + (int64_t)row * p.ldc + gcol) = v;
                } else {
                    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *(u32x2_t*)((uint16_t*)p.C + (int64_t)row * p.ldc + gcol) = o;
                }
            }
        }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    if (p.res_dtype == FW_DT_F32) {
        pass(Q0{}, std::integral_constant<int, FW_DT_F32>{});
        pass(Q1{}, std::integral_constant<int, FW_DT_F32>{});
    } else if (p.res_dtype == FW_DT_BF16) {
        pass(Q0{}, std::integral_constant<int, FW_DT_BF16>{});
        pass(Q1{}, std::integral_constant<int, FW_DT_BF16>{});
    } else {
        pass(Q0{}, std::integral_constant<int, FW_DT_NONE>{});
        pass(Q1{}, std::integral_constant<int, FW_DT_NONE>{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 accumulators of 32x32.  One work-group per CU, so the
// two waves that share a SIMD belong to the SAME work-group; to keep the matrix pipe fed they are run half a phase
// apart: waves 0-3 (group A) and 4-7 (group B) execute the same sequence of  [LOAD fragments | barrier | 16 MFMA |
// barrier]  phases, but group B takes one extra barrier at the start (and group A one at the end), so that while one
// wave of a SIMD is in its MFMA segment its partner is in its LDS-read segment.  A k-slab is consumed in two phases
// (wave rows 0-63, then 64-127; the B fragments are read once per slab).  The next slab is streamed with
// global_load_lds during the phases in which its LDS stage is provably idle, and waited for (vmcnt(0)) one barrier
// before its first read:
//     global slot:      4t      4t+1      4t+2      4t+3      4t+4
//     group A:        LOAD(t,0) MFMA(t,0) LOAD(t,1) MFMA(t,1) LOAD(t+1,0)
//     group B:        MFMA(..)  LOAD(t,0) MFMA(t,0) LOAD(t,1) MFMA(t,1)
//     slab t+1 DMA:   A: 1/2    B: all    A: 1/2    wait      first read (A)   -- last read of slab t-1 completes
//                                                                                  before the barrier ending slot 4t-1
// Barriers are raw s_barrier (inline asm): a __syncthreads() would drain the DMA queue at every barrier.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TM = 256, TN = 256;
constexpr int STAGE2 = (TM + TN) * BK * 2;   // 64 KiB

#define FW_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FW_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(GemmArgs p) {
    constexpr bool DMA_IN_LOAD = (VAR & 1) != 0, DRAIN_LDS = (VAR & 2) != 0, PRIO = (VAR & 4) != 0;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = (VAR & 8) ? (wave & 1) : (wave >> 2);
    const int wn = (VAR & 8) ? (wave >> 1) : (wave & 3);

    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tm, tn;
    {
        const int per_group = GROUP_M * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    // staging: wave w streams A pieces 4w..4w+3 and W pieces 4w..4w+3 of every slab (1 KiB = 8 rows x 128 B each).
    // Uniform 64-bit tile base (SGPR) + per-lane 32-bit element offsets keep the address state at 8 VGPRs.
    const uint16_t* abase = p.A + (int64_t)m0 * p.lda;
    const uint16_t* wbase = p.W + (int64_t)n0 * p.ldw;
    int aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        aoff[i] = min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8;
        woff[i] = min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8;
    }
#define FW_STAGE_PIECE(S, KT, I)                                                                   \
    do {                                                                                           \
        FW_GLDS16(abase + (KT) * BK + aoff[I], smem + (S) * STAGE2 + (wave * 4 + (I)) * 1024);     \
        FW_GLDS16(wbase + (KT) * BK + woff[I], smem + (S) * STAGE2 + TM * BK * 2 + (wave * 4 + (I)) * 1024); \
    } while (0)
#define FW_STAGE_FIRST(S, KT) do { FW_STAGE_PIECE(S, KT, 0); FW_STAGE_PIECE(S, KT, 1); } while (0)
#define FW_STAGE_SECOND(S, KT) do { FW_STAGE_PIECE(S, KT, 2); FW_STAGE_PIECE(S, KT, 3); } while (0)

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;
    const int a_row_off = (grp * 128 + fi) * 128;                    // + rb*32*128, rb = 0..3
    const int b_row_off = TM * BK * 2 + (wn * 64 + fi) * 128;        // + nb*32*128

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t afr[2][4], bfr[2][4];

    const int nk = p.K / BK;
    FW_STAGE_FIRST(0, 0);
    FW_STAGE_SECOND(0, 0);
    FW_WAIT_DMA();
    FW_BARRIER();
    if (grp == 1) FW_BARRIER();

    for (int kt = 0; kt < nk; ++kt) {
        const char* base = smem + (kt & 1) * STAGE2;
        const bool more = kt + 1 < nk;
        const int ns = (kt + 1) & 1;
        // ---------------- phase (kt, 0): B fragments + A rows 0..63 of the wave tile.
        // DMA of slab kt+1 is issued only from LOAD segments (never in front of an MFMA burst): group A issues half in
        // each of its two LOAD segments (slots 4t, 4t+2), group B all of it in LOAD(kt,0) (slot 4t+1).  Every LOAD
        // segment drains its own ds_reads (lgkmcnt(0)) BEFORE the barrier, so when a barrier releases, no read of the
        // previous slab is in flight and the stage may be overwritten from slot 4t on.
        if (more) {
            if (DMA_IN_LOAD) {
                FW_STAGE_FIRST(ns, kt + 1);
                if (grp == 1) FW_STAGE_SECOND(ns, kt + 1);
            } else if (grp == 1) {
                FW_STAGE_FIRST(ns, kt + 1);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bfr[0][ks] = *(const bf16x8_t*)(base + b_row_off + coff[ks]);
            bfr[1][ks] = *(const bf16x8_t*)(base + b_row_off + 32 * 128 + coff[ks]);
            afr[0][ks] = *(const bf16x8_t*)(base + a_row_off + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(base + a_row_off + 32 * 128 + coff[ks]);
        }
        if (DRAIN_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FW_BARRIER();
        if (!DMA_IN_LOAD && more) {
            if (grp == 0) FW_STAGE_FIRST(ns, kt + 1);
            else FW_STAGE_SECOND(ns, kt + 1);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[1][1], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        FW_BARRIER();
        // ---------------- phase (kt, 1): A rows 64..127 of the wave tile
        if (grp == 0 && more) FW_STAGE_SECOND(ns, kt + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            afr[0][ks] = *(const bf16x8_t*)(base + a_row_off + 64 * 128 + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(base + a_row_off + 96 * 128 + coff[ks]);
        }
        if (DRAIN_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) FW_WAIT_DMA();          // group B: end of slot 4t+3
        FW_BARRIER();
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[2][1], 0, 0, 0);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[3][1], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (grp == 0) FW_WAIT_DMA();          // group A: end of slot 4t+3
        FW_BARRIER();
    }
    if (grp == 0) FW_BARRIER();

    epilogue_256(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}


// ---------------------------------------------------------------------------------------------------------------
// 256x256 tile, k streamed in HALF-SLABS of 32 through a 5-deep LDS ring (5 x 32 KiB = all 160 KiB of the CU).
//
// Why: with two 64 KiB stages a slab's DMA has at most ~1 slab time to land and every slab ends in vmcnt(0); a
// global_load_lds that misses L2 needs ~1-2.5k cycles under load, so the matrix pipe idles on the slowest piece of every
// slab.  Here a half-slab is requested FOUR half-slabs (8 barrier slots) before its first read and retired with a COUNTED
// vmcnt (12 / 8 pieces still in flight), so the loop never drains the DMA queue.
//
// 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 accumulators of 32x32 (v_mfma_f32_32x32x16_bf16); per half-slab a wave
// does 12 ds_read_b128 (A 4x2, B 2x2) and 16 MFMA.  The two 4-wave groups (waves sharing a SIMD are in different groups)
// run one barrier slot apart, so one is in its MFMA burst while its SIMD partner reads LDS:
//     slot:        2h          2h+1        2h+2
//     group A:   LOAD(h)     MFMA(h)     LOAD(h+1)
//     group B:   MFMA(h-1)   LOAD(h)     MFMA(h)
// Ring slot h%5 is last read in slot 2h+1 (reads drained with lgkmcnt(0) before the barrier), and rewritten with half-slab
// h+5-1 = h+4's successor: every wave issues its 4 pieces of half-slab h+4 from INSIDE MFMA(h) (slot >= 2h+1 > 2(h-1)+1),
// interleaved with the MFMAs so the issue cost hides behind the matrix pipe.  Half-slab j must be complete before group A's
// LOAD(j) in slot 2j: group A waits vmcnt(12) at the end of MFMA(j-1) (younger: j+1..j+3), group B waits vmcnt(8) at the end
// of LOAD(j-1) (younger: j+1, j+2); both waits precede the barrier that ends slot 2j-1.
// LDS image of a half-slab: 512 rows (256 A, 256 W) x 64 B, 16-B chunk index XOR ((row>>2)&3) -- applied to the per-lane
// SOURCE address of the DMA (destination is lane-linear) and again on the read: conflict-free ds_read_b128.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HK = 32;                          // k per half-slab
constexpr int RING = 5;
constexpr int HSLAB = (TM + TN) * HK * 2;       // 32 KiB

template <int N> __device__ __forceinline__ void fw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm_bf16_ring_kernel(GemmArgs p) {
    constexpr bool PRIO = (VAR & 1) != 0;        // s_setprio(1) around the MFMA burst
    constexpr bool DMA_HEAD = (VAR & 2) != 0;    // issue the 4 DMA pieces in front of the burst instead of inside it
    constexpr bool NO_DMA = (VAR & 4) != 0;      // ablation: never restage (wrong results; measures the LDS/MFMA/barrier loop)
    constexpr bool DMA_LOAD = (VAR & 8) != 0;    // issue the 4 DMA pieces from the LOAD slot (while the SIMD partner owns the matrix pipe)
    constexpr bool ROW128 = (VAR & 64) != 0;     // ablation (wrong results): every DMA instruction reads 8 rows x 128 B instead of 16 rows x 64 B
    constexpr bool FEW_LANES = (VAR & 32) != 0;  // ablation (wrong results): only 4 lanes of every DMA instruction active (same instruction count, 1/16 of the bytes)
    constexpr bool SAME_ADDR = (VAR & 16) != 0;  // ablation (wrong results): every DMA re-reads half-slab 0 (L2-hot) -- memory system vs CU-internal cost
    __shared__ __attribute__((aligned(16))) char smem[RING * HSLAB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wn = wave & 3;

    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tm, tn;
    {
        const int per_group = GROUP_M * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    // ---- DMA addressing: wave w streams A pieces 2w, 2w+1 and W pieces 2w, 2w+1 of every half-slab (1 KiB = 16 rows x 64 B).
    // Uniform 64-bit base (SGPR pair, advanced by 64 B per half-slab) + per-lane unsigned 32-bit byte offset (VGPR): the
    // saddr form of global_load_lds, no per-piece 64-bit VALU address arithmetic.
    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[2], woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = (wave * 2 + i) * 16 + (lane >> 2);
        int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if (ROW128) { row = (wave * 2 + i) * 16 + (lane >> 3); chunk = lane & 7; }
        aoff[i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
        woff[i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
    }
#define FW_RING_PIECE_A(RS, H, I) if (!FEW_LANES || lane < 4) FW_GLDS16(abase + (size_t)(SAME_ADDR ? 0 : (H)) * (HK * 2) + aoff[I], smem + (RS) * HSLAB + (wave * 2 + (I)) * 1024)
#define FW_RING_PIECE_W(RS, H, I) if (!FEW_LANES || lane < 4) FW_GLDS16(wbase + (size_t)(SAME_ADDR ? 0 : (H)) * (HK * 2) + woff[I], smem + (RS) * HSLAB + TM * HK * 2 + (wave * 2 + (I)) * 1024)

    // ---- fragment read offsets: row R, logical chunk c = 2*ks + hi, physical chunk = c ^ ((R>>2)&3) ---------------------
    const int fi = lane & 31, hi = lane >> 5;
    const int f = (fi >> 2) & 3;
    const int a_off0 = (grp * 128 + fi) * 64 + (((0 + hi) ^ f) << 4);
    const int a_off1 = (grp * 128 + fi) * 64 + (((2 + hi) ^ f) << 4);
    const int b_off0 = TM * HK * 2 + (wn * 64 + fi) * 64 + (((0 + hi) ^ f) << 4);
    const int b_off1 = TM * HK * 2 + (wn * 64 + fi) * 64 + (((2 + hi) ^ f) << 4);

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t afr[4][2], bfr[2][2];

    const int nh = p.K / HK;        // >= 8 (launcher)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        FW_RING_PIECE_A(h, h, 0); FW_RING_PIECE_W(h, h, 0); FW_RING_PIECE_A(h, h, 1); FW_RING_PIECE_W(h, h, 1);
    }
    fw_wait_vm<12>();
    FW_BARRIER();
    if (grp == 1) FW_BARRIER();

    int rs = 0;                      // ring slot of half-slab h
    int h = 0;
    // one LOAD(h) | barrier | MFMA(h) (+ DMA of half-slab h+4 when MORE) | barrier
    auto iter = [&](auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value && !NO_DMA;
        // ------------------------------------------------ LOAD(h)
        {
            const char* base = smem + rs * HSLAB;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                bfr[nb][0] = *(const bf16x8_t*)(base + b_off0 + nb * 2048);
                bfr[nb][1] = *(const bf16x8_t*)(base + b_off1 + nb * 2048);
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                afr[rb][0] = *(const bf16x8_t*)(base + a_off0 + rb * 2048);
                afr[rb][1] = *(const bf16x8_t*)(base + a_off1 + rb * 2048);
            }
        }
        const int ns = rs == 0 ? 4 : rs - 1;
        if (DMA_LOAD && MORE) {
            // ring slot ns held half-slab h-1, last read in slot 2h-1 (group B's LOAD(h-1)); this LOAD runs in slot >= 2h
            FW_RING_PIECE_A(ns, h + 4, 0); FW_RING_PIECE_W(ns, h + 4, 0); FW_RING_PIECE_A(ns, h + 4, 1); FW_RING_PIECE_W(ns, h + 4, 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) {
            if (!decltype(more_tag)::value) fw_wait_vm<0>();
            else if (DMA_LOAD) fw_wait_vm<12>();      // younger than half-slab h+1: h+2, h+3, h+4
            else fw_wait_vm<8>();                     // younger than half-slab h+1: h+2, h+3
        }
        FW_BARRIER();
        // ------------------------------------------------ MFMA(h), DMA of half-slab h+4 into ring slot (rs+4)%5
        if (DMA_HEAD && !DMA_LOAD && MORE) {
            FW_RING_PIECE_A(ns, h + 4, 0); FW_RING_PIECE_W(ns, h + 4, 0); FW_RING_PIECE_A(ns, h + 4, 1); FW_RING_PIECE_W(ns, h + 4, 1);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[rb][ks], bfr[0][ks], acc[rb][0], 0, 0, 0);
                acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[rb][ks], bfr[1][ks], acc[rb][1], 0, 0, 0);
                if (!DMA_HEAD && !DMA_LOAD && MORE) {
                    // one DMA piece after MFMAs 2, 6, 10, 14 of the burst
                    if (ks == 0 && rb == 0) FW_RING_PIECE_A(ns, h + 4, 0);
                    if (ks == 0 && rb == 2) FW_RING_PIECE_W(ns, h + 4, 0);
                    if (ks == 1 && rb == 0) FW_RING_PIECE_A(ns, h + 4, 1);
                    if (ks == 1 && rb == 2) FW_RING_PIECE_W(ns, h + 4, 1);
                }
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (grp == 0) { if (decltype(more_tag)::value) fw_wait_vm<12>(); else fw_wait_vm<0>(); }
        FW_BARRIER();
        rs = rs == RING - 1 ? 0 : rs + 1;
    };
    for (; h < nh - 4; ++h) iter(std::true_type{});
    for (; h < nh; ++h) iter(std::false_type{});
    if (grp == 0) FW_BARRIER();
    epilogue_256(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}


// ---------------------------------------------------------------------------------------------------------------
// 256x256 tile with FOUR waves (256 threads), one wave per SIMD owning the whole register file: wave tile 128 x 128 =
// 16 accumulators of 32x32 (256 accumulator registers, placed in the AGPR half by the compiler) + fragments / addresses
// in the 256 architectural VGPRs.  No inter-wave ping-pong: every wave software-pipelines ITS OWN stream -- fragment
// ds_reads of the next k-step and the LDS-DMA of a later half-slab are issued between the MFMAs of the current k-step
// (32 MFMAs per 32-wide half-slab, one ds_read_b128 per 2 MFMAs, one DMA piece per 4), so the matrix pipe is fed by a
// single in-order instruction stream and there is only ONE barrier per half-slab (DMA visibility + ring reuse).
// LDS traffic per slab drops to 2/3 of the 8-wave kernels (wave tile 128x128 instead of 128x64).
// Ring: the same 5 x 32 KiB half-slab ring and swizzle as gemm_bf16_ring_kernel.  In iteration h (compute half-slab h) the
// wave issues its 8 pieces of half-slab h+4 into the slot of half-slab h-1 (all reads of h-1 retired before the barrier
// of iteration h-1) and, before the barrier that opens half-slab h+1, waits vmcnt(20): younger than its pieces of h+1 are
// h+2, h+3 (16) and the first 4 pieces of h+4.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void epilogue_w4(const GemmArgs& p, char* smem, f32x16_t (&acc)[4][4], int wave, int wm, int wn,
                                            int fi, int hi, int lane, int m0, int n0) {
    // Pass rb: the wave's 32 x 128 fp32 block goes through a private 16 KiB LDS region (raw accumulators in, row-contiguous
    // 16 B per lane out); bias / activation / per-column affine / residual are applied on the way out, where a lane owns 4
    // fixed columns and whole 512-B (fp32) / 256-B (bf16) row segments are read and written.
    char* reg = smem + wave * 16384;
    const int rl = lane >> 5;                  // row inside a 2-row read group
    const int c4 = (lane & 31) * 4;            // first of this lane's 4 columns
    const int gcol = n0 + wn * 128 + c4;
    const bool col_ok = gcol < p.N;            // N % 4 == 0 (launcher): the 4 columns are valid together
    f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f}, g14 = {1.f, 1.f, 1.f, 1.f}, g04 = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
        if (p.bias) bias4 = *(const f32x4_t*)(p.bias + gcol);
        if (p.g1) g14 = *(const f32x4_t*)(p.g1 + gcol);
        if (p.g0) g04 = *(const f32x4_t*)(p.g0 + gcol);
    }
    const int act = p.act;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_l = (r & 3) + 8 * (r >> 2) + 4 * hi;
                *(float*)(reg + row_l * 512 + (nb * 32 + fi) * 4) = acc[rb][nb][r];
            }
#pragma unroll 2
        for (int it = 0; it < 16; ++it) {
            const int row_l = it * 2 + rl;
            f32x4_t v = *(const f32x4_t*)(reg + row_l * 512 + c4 * 4);
            const int row = m0 + wm * 128 + rb * 32 + row_l;
            if (row < p.M && col_ok) {
                v += bias4;
                if (act != FW_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fw_apply_act(v[j], act);
                }
                v = v * g14 + g04;
                if (p.res_dtype == FW_DT_F32) {
                    const f32x4_t rv = *(const f32x4_t*)((const float*)p.res + (int64_t)row * p.ldr + gcol);
                    v += rv;
                } else if (p.res_dtype == FW_DT_BF16) {
                    const u32x2_t rw = *(const u32x2_t*)((const uint16_t*)p.res + (int64_t)row * p.ldr + gcol);
                    v[0] += __uint_as_float(rw[0] << 16); v[1] += __uint_as_float(rw[0] & 0xffff0000u);
                    v[2] += __uint_as_float(rw[1] << 16); v[3] += __uint_as_float(rw[1] & 0xffff0000u);
                }
                if (p.out_dtype == FW_DT_F32) {
                    *(f32x4_t*)((float*)p.C + (int64_t)row * p.ldc + gcol) = v;
                } else {
                    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *(u32x2_t*)((uint16_t*)p.C + (int64_t)row * p.ldc + gcol) = o;
                }
            }
        }
    }
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(GemmArgs p) {
    constexpr bool PRIO = (VAR & 1) != 0;
    constexpr bool NO_DMA = (VAR & 4) != 0;      // ablation (wrong results)
    __shared__ __attribute__((aligned(16))) char smem[RING * HSLAB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tm, tn;
    {
        const int per_group = GROUP_M * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    // DMA: wave w streams A pieces 4w..4w+3 and W pieces 4w..4w+3 of every half-slab (1 KiB = 16 rows x 64 B)
    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        aoff[i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
        woff[i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
    }
#define FW_W4_PIECE_A(RS, H, I) FW_GLDS16(abase + (size_t)(H) * (HK * 2) + aoff[I], smem + (RS) * HSLAB + (wave * 4 + (I)) * 1024)
#define FW_W4_PIECE_W(RS, H, I) FW_GLDS16(wbase + (size_t)(H) * (HK * 2) + woff[I], smem + (RS) * HSLAB + TM * HK * 2 + (wave * 4 + (I)) * 1024)

    const int fi = lane & 31, hi = lane >> 5;
    const int f = (fi >> 2) & 3;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_off[ks] = (wm * 128 + fi) * 64 + (((2 * ks + hi) ^ f) << 4);
        b_off[ks] = TM * HK * 2 + (wn * 128 + fi) * 64 + (((2 * ks + hi) ^ f) << 4);
    }

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t a0[4], b0[4], a1[4], b1[4];         // fragment sets of k-step 0 / 1 of a half-slab

    const int nh = p.K / HK;        // >= 8 (launcher)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { FW_W4_PIECE_A(h, h, i); FW_W4_PIECE_W(h, h, i); }
    }
    fw_wait_vm<24>();
    FW_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a0[i] = *(const bf16x8_t*)(smem + a_off[0] + i * 2048);
        b0[i] = *(const bf16x8_t*)(smem + b_off[0] + i * 2048);
    }

    int rs = 0;
    int h = 0;
    auto iter = [&](auto more_tag, auto last_tag) {
        constexpr bool MORE = decltype(more_tag)::value && !NO_DMA;     // half-slab h+4 exists
        constexpr bool LAST = decltype(last_tag)::value;                // h == nh-1
        const char* base = smem + rs * HSLAB;
        const int ns = rs == 0 ? 4 : rs - 1;       // ring slot of half-slab h+4 (== h-1)
        const int nx = rs == RING - 1 ? 0 : rs + 1;
        // ---- k-step 0: prefetch k-step 1 fragments, first half of the DMA, 16 MFMAs on set 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a1[i] = *(const bf16x8_t*)(base + a_off[1] + i * 2048);
            b1[i] = *(const bf16x8_t*)(base + b_off[1] + i * 2048);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[rb], b0[nb], acc[rb][nb], 0, 0, 0);
            if (MORE) { if (rb & 1) FW_W4_PIECE_W(ns, h + 4, rb >> 1); else FW_W4_PIECE_A(ns, h + 4, rb >> 1); }
        }
        // ---- k-step 1: open half-slab h+1 (own pieces landed -> barrier), prefetch its k-step 0, rest of the DMA, 16 MFMAs
        if (!LAST) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MORE) fw_wait_vm<20>(); else fw_wait_vm<0>();
            FW_BARRIER();
            const char* nbase = smem + nx * HSLAB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0[i] = *(const bf16x8_t*)(nbase + a_off[0] + i * 2048);
                b0[i] = *(const bf16x8_t*)(nbase + b_off[0] + i * 2048);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[rb], b1[nb], acc[rb][nb], 0, 0, 0);
            if (MORE) { if (rb & 1) FW_W4_PIECE_W(ns, h + 4, 2 + (rb >> 1)); else FW_W4_PIECE_A(ns, h + 4, 2 + (rb >> 1)); }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        rs = nx;
    };
    for (; h < nh - 4; ++h) iter(std::true_type{}, std::false_type{});
    for (; h < nh - 1; ++h) iter(std::false_type{}, std::false_type{});
    iter(std::false_type{}, std::true_type{});
    FW_BARRIER();
    epilogue_w4(p, smem, acc, wave, wm, wn, fi, hi, lane, m0, n0);
}


// ---------------------------------------------------------------------------------------------------------------
// 256x256x64 ping-pong kernel, 128-B LDS rows, two 64 KiB stages, quarter-slab DMA scheduling with counted waits.
//
// Measured on MI355X (tools/probes/dma_probe.hip): the LDS-DMA ingest of a CU tops out at ~107 GB/s with 128-B global
// rows but only ~67 GB/s with 64-B rows (57 vs 85 GB/s beside ds_read traffic), and the GEMM's cost scales with the BYTES
// it streams, not with the number of DMA instructions.  So this kernel keeps full 128-B rows (k-slab 64) like
// gemm_bf16_256_kernel, and gets its prefetch distance from scheduling instead of from more LDS: a slab is four 16 KiB
// units -- A0 = tile rows 0..127 (read only by wave group A), A1 = rows 128..255 (group B), B0/B1 = the W rows -- and a
// unit of slab t+2 is requested as soon as the same unit of slab t has been read for the last time:
//     slot:        4t        4t+1      4t+2      4t+3      4t+4
//     group A:   LOAD0(t)  MFMA0(t)  LOAD1(t)  MFMA1(t)  LOAD0(t+1)      LOAD0 reads B(t) + A rows 0-63 of the wave tile,
//     group B:   MFMA1(..) LOAD0(t)  MFMA0(t)  LOAD1(t)  MFMA1(t)        LOAD1 reads A rows 64-127
//   B(t) is last read in slot 4t+1, A0(t) in 4t+2, A1(t) in 4t+3.  Every wave issues, from inside its MFMA1(t) burst (slot
//   4t+3 / 4t+4), its 6 pieces of B(t+2) and A0(t+2), and from MFMA0(t+1) its 2 pieces of A1(t+2): 4-5 slots (~one slab
//   time) before the first read in slot 4t+8 / 4t+9.  Waits are counted (never vmcnt(0) in steady state):
//     group A: end of MFMA1(t): vmcnt(8) -> own B/A0(t+1) landed;   end of LOAD0(t): vmcnt(6) -> own A1(t) landed
//     group B: end of LOAD1(t): vmcnt(2) -> own B/A0(t+1) landed;   end of MFMA1(t): vmcnt(6) -> own A1(t+1) landed
//   each ahead of the barrier that precedes the first read of that unit by any wave.
// ---------------------------------------------------------------------------------------------------------------
template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(GemmArgs p) {
    constexpr bool PRIO = (VAR & 1) != 0;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wn = wave & 3;

    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tm, tn;
    {
        const int per_group = GROUP_M * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    // DMA pieces (1 KiB = 8 rows x 128 B).  Wave w owns, in every 128-row unit, pieces 2w and 2w+1 (rows 16w .. 16w+15).
    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[2][2], woff[2][2];       // [unit][piece]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = u * 128 + (wave * 2 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            aoff[u][i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
            woff[u][i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
        }
#define FW_PP_A(S, KT, U, I) FW_GLDS16(abase + (size_t)(KT) * (BK * 2) + aoff[U][I], smem + (S) * STAGE2 + ((U) * 16 + wave * 2 + (I)) * 1024)
#define FW_PP_W(S, KT, U, I) FW_GLDS16(wbase + (size_t)(KT) * (BK * 2) + woff[U][I], smem + (S) * STAGE2 + TM * BK * 2 + ((U) * 16 + wave * 2 + (I)) * 1024)
#define FW_PP_ISSUE6(S, KT) do { FW_PP_W(S, KT, 0, 0); FW_PP_W(S, KT, 0, 1); FW_PP_W(S, KT, 1, 0); FW_PP_W(S, KT, 1, 1); FW_PP_A(S, KT, 0, 0); FW_PP_A(S, KT, 0, 1); } while (0)
#define FW_PP_ISSUE2(S, KT) do { FW_PP_A(S, KT, 1, 0); FW_PP_A(S, KT, 1, 1); } while (0)

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;
    const int a_row_off = (grp * 128 + fi) * 128;                    // + rb*32*128, rb = 0..3
    const int b_row_off = TM * BK * 2 + (wn * 64 + fi) * 128;        // + nb*32*128

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t afr[2][4], bfr[2][4];

    const int nk = p.K / BK;       // >= 4 (launcher)
    FW_PP_ISSUE6(0, 0); FW_PP_ISSUE2(0, 0);
    FW_PP_ISSUE6(1, 1); FW_PP_ISSUE2(1, 1);
    fw_wait_vm<8>();
    FW_BARRIER();
    if (grp == 1) FW_BARRIER();

    for (int kt = 0; kt < nk; ++kt) {
        const char* base = smem + (kt & 1) * STAGE2;
        const int st = kt & 1;
        const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
        // ---------------- LOAD0(kt): B fragments + A rows 0..63 of the wave tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bfr[0][ks] = *(const bf16x8_t*)(base + b_row_off + coff[ks]);
            bfr[1][ks] = *(const bf16x8_t*)(base + b_row_off + 32 * 128 + coff[ks]);
            afr[0][ks] = *(const bf16x8_t*)(base + a_row_off + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(base + a_row_off + 32 * 128 + coff[ks]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 0) { if (has1) fw_wait_vm<6>(); else fw_wait_vm<0>(); }
        FW_BARRIER();
        // ---------------- MFMA0(kt) (+ A1 unit of slab kt+1 into the other stage; slabs 0 and 1 come from the prologue)
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[0][1], 0, 0, 0);
            if (ks == 0 && kt >= 1 && has1) FW_PP_ISSUE2(st ^ 1, kt + 1);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[1][1], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        FW_BARRIER();
        // ---------------- LOAD1(kt): A rows 64..127 of the wave tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            afr[0][ks] = *(const bf16x8_t*)(base + a_row_off + 64 * 128 + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(base + a_row_off + 96 * 128 + coff[ks]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) { if (has1) fw_wait_vm<2>(); else fw_wait_vm<0>(); }
        FW_BARRIER();
        // ---------------- MFMA1(kt) (+ B and A0 units of slab kt+2 into this stage)
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[2][1], 0, 0, 0);
            if (ks == 0 && has2) FW_PP_ISSUE6(st, kt + 2);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[3][1], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (grp == 0) { if (has2) fw_wait_vm<8>(); else if (has1) fw_wait_vm<2>(); else fw_wait_vm<0>(); }
        else { if (has2) fw_wait_vm<6>(); else fw_wait_vm<0>(); }
        FW_BARRIER();
    }
    if (grp == 0) FW_BARRIER();
    epilogue_256(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// Second generation of the ping-pong kernel (round 2, the default).  Same tile, LDS image, slot table, DMA placement, counted waits
// and fragment reads as gemm_bf16_pp_kernel; what changed was read off the ISA:
//   * the loop is peeled by hand (first / steady / second-last / last slab as compile-time flags), so a phase is ONE basic block;
//   * the accumulators are pinned by an empty asm after every burst, so no MFMA drifts across the barrier that follows.
// +2..5 % on the DiT shapes (profiles/r02/gemm_experiments.md).  Tried on this skeleton and measured WITHOUT gain, hence not kept:
// LDS-DMA issued from the LOAD phases instead of from inside the bursts (+-0), the slot barrier signalled one k-step early so its
// release latency runs under the last MFMAs (+-0), a software L2 prefetch 3-4 slabs ahead (-10 %: the extra line requests cost
// more than the misses they hide), a rotated K start per work-group against channel hot-spotting (-6 %: it breaks the lockstep L2
// sharing of A bands / W panels), staggered work-group starts to de-phase the residual epilogues (-3..-20 %).  The TIMING build
// (TS, tools/gemm_timeline.py) stamps s_memtime at the phase boundaries.
// ---------------------------------------------------------------------------------------------------------------
__device__ unsigned long long g_gemm_ts[2 * 64];     // TIMING build: [group][slab 0..3][phase 0..3][start | end of work]

template <bool TS>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp2_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wn = wave & 3;

    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tm, tn;
    {
        const int per_group = GROUP_M * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[2][2], woff[2][2];       // [unit][piece]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = u * 128 + (wave * 2 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            aoff[u][i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
            woff[u][i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
        }

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;
    const int a_row_off = (grp * 128 + fi) * 128;
    const int b_row_off = TM * BK * 2 + (wn * 64 + fi) * 128;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t afr[2][4], bfr[2][4];

    const int nk = p.K / BK;       // >= 4 (launcher)
    FW_PP_ISSUE6(0, 0); FW_PP_ISSUE2(0, 0);
    FW_PP_ISSUE6(1, 1); FW_PP_ISSUE2(1, 1);
    fw_wait_vm<8>();
    FW_BARRIER();
    if (grp == 1) FW_BARRIER();

    int kt = 0;
    auto slab = [&](auto first_tag, auto has1_tag, auto has2_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, HAS1 = decltype(has1_tag)::value, HAS2 = decltype(has2_tag)::value;
        const int st = kt & 1;
        const char* base = smem + st * STAGE2;
        // TIMING build: s_memtime at the start (barrier passed) and at the end of the work of every phase, slabs 16..19, work-group 0
        const bool ts_on = TS && blockIdx.x == 0 && wn == 0 && kt >= 16 && kt < 20;
        auto stamp = [&](int phase, int which) {
            if (ts_on) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) g_gemm_ts[grp * 64 + (kt - 16) * 8 + phase * 2 + which] = t;
            }
        };
        stamp(0, 0);
        // ---------------- LOAD0(kt): B fragments + A rows 0..63 of the wave tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bfr[0][ks] = *(const bf16x8_t*)(base + b_row_off + coff[ks]);
            bfr[1][ks] = *(const bf16x8_t*)(base + b_row_off + 32 * 128 + coff[ks]);
            afr[0][ks] = *(const bf16x8_t*)(base + a_row_off + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(base + a_row_off + 32 * 128 + coff[ks]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 0) { if (HAS1) fw_wait_vm<6>(); else fw_wait_vm<0>(); }
        stamp(0, 1);
        FW_BARRIER();
        stamp(1, 0);
        // ---------------- MFMA0(kt) (+ A1 unit of slab kt+1 into the other stage; slabs 0 and 1 come from the prologue)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[0][1], 0, 0, 0);
            if (ks == 0 && !FIRST && HAS1) FW_PP_ISSUE2(st ^ 1, kt + 1);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        stamp(1, 1);
        FW_BARRIER();
        stamp(2, 0);
        // ---------------- LOAD1(kt): A rows 64..127 of the wave tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            afr[0][ks] = *(const bf16x8_t*)(base + a_row_off + 64 * 128 + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(base + a_row_off + 96 * 128 + coff[ks]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) { if (HAS1) fw_wait_vm<2>(); else fw_wait_vm<0>(); }
        stamp(2, 1);
        FW_BARRIER();
        stamp(3, 0);
        // ---------------- MFMA1(kt) (+ B and A0 units of slab kt+2 into this stage)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[2][1], 0, 0, 0);
            if (ks == 0 && HAS2) FW_PP_ISSUE6(st, kt + 2);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[3][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        stamp(3, 1);
        if (grp == 0) { if (HAS2) fw_wait_vm<8>(); else if (HAS1) fw_wait_vm<2>(); else fw_wait_vm<0>(); }
        else { if (HAS2) fw_wait_vm<6>(); else fw_wait_vm<0>(); }
        FW_BARRIER();
        ++kt;
    };
    using T = std::true_type;
    using F = std::false_type;
    slab(T{}, T{}, T{});
    while (kt < nk - 2) slab(F{}, T{}, T{});
    slab(F{}, T{}, F{});
    slab(F{}, F{}, F{});
    if (grp == 0) FW_BARRIER();
    epilogue_256(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}

// fp32 GEMV for the M=1 time-embedding MLPs: one wave per output feature.
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ x, const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int N, int K, int act_in_silu, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (int64_t)n * ldw;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = x[k];
        if (act_in_silu) xv = fw_silu(xv);
        s += xv * w[k];
    }
    s = wave_sum(s);
    if (lane == 0) {
        s += bias ? bias[n] : 0.f;
        out[n] = fw_apply_act(s, act_out);
    }
}

}  // namespace

extern "C" int fw_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                            void* C, int64_t ldc, int out_dtype, int M, int N, int K,
                            const float* bias, int act, const float* g1, const float* g0,
                            const void* res, int64_t ldr, int res_dtype, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K % BK) != 0) { fw_set_error("fw_gemm_bf16: K must be a positive multiple of 64"); return FW_E_BADARG; }
    if ((lda % 8) || (ldw % 8) || (((uintptr_t)A) & 15) || (((uintptr_t)W) & 15)) {
        fw_set_error("fw_gemm_bf16: A/W must be 16-byte aligned with lda/ldw % 8 == 0"); return FW_E_BADARG; }
    if (out_dtype != FW_DT_BF16 && out_dtype != FW_DT_F32) { fw_set_error("fw_gemm_bf16: bad out_dtype"); return FW_E_BADARG; }
    if (res_dtype != FW_DT_NONE && res == nullptr) { fw_set_error("fw_gemm_bf16: res_dtype set but res NULL"); return FW_E_BADARG; }
    GemmArgs p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.out_dtype = out_dtype;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.act = act; p.g1 = g1; p.g0 = g0;
    p.res = res; p.ldr = ldr; p.res_dtype = res ? res_dtype : FW_DT_NONE;
    // tile choice: the 256x256 staggered kernel for the big token-major GEMMs, 128x128 otherwise.
    // FW_GEMM_TILE=128|256 forces one (A/B measurements).
    const int forced = fw_get_option(FW_OPT_GEMM_TILE);
    // (narrow outputs qualify once there are two full rounds of 256x256 tiles anyway: the geometry heads' convolutions are
    // GEMMs with N = 256..512 over hundreds of thousands of pixel rows)
    bool big = M >= 2048 && (N >= 1024 || (N >= 256 && (int64_t)(M / TM) * ((N + TN - 1) / TN) >= 512));
    if (forced == 128) big = false;
    if (forced == 256) big = true;
    // the 256 kernel's epilogue moves 4 columns per lane: needs N, ldc, ldr % 4 == 0 and 16-B (fp32) / 8-B (bf16) bases
    const uintptr_t cmask = (out_dtype == FW_DT_F32) ? 15 : 7;
    const uintptr_t rmask = (res_dtype == FW_DT_F32) ? 15 : 7;
    if ((N % 4) || (ldc % 4) || (((uintptr_t)C) & cmask) || (res && ((ldr % 4) || (((uintptr_t)res) & rmask)))) big = false;
    if ((((uintptr_t)bias) | ((uintptr_t)g1) | ((uintptr_t)g0)) & 15) big = false;   // per-column vectors are read 16 B at a time
    if (big) {
        const int kern = fw_get_option(FW_OPT_GEMM_KERNEL);
        // Short M tail (VGGT: 32865 rows = 128 full row bands + 97 rows): a 129th band of 256-row tiles costs a whole extra
        // round of the grid (516 tiles on 256 CUs = 3 rounds for 2.02 rounds of work).  Peel it: the full bands go to the 256x256
        // kernel (512 tiles = 2 rounds), the <= 128 leftover rows to the 128x128 kernel in a second, tiny launch.  Same
        // k-order per output element in both kernels, disjoint output rows.
        const int tail = M % TM;
        if ((kern == 3 || kern == 4) && tail > 0 && tail <= BM && M >= 2 * TM) {
            const int Mfull = M - tail;
            const size_t cbytes = (out_dtype == FW_DT_F32) ? 4 : 2, rbytes = (res_dtype == FW_DT_F32) ? 4 : 2;
            int rc = fw_gemm_bf16(A, lda, W, ldw, C, ldc, out_dtype, Mfull, N, K, bias, act, g1, g0, res, ldr, res_dtype, stream);
            if (rc) return rc;
            GemmArgs t = p;
            t.A = A + (int64_t)Mfull * lda;
            t.C = (char*)C + (size_t)Mfull * ldc * cbytes;
            t.res = res ? (const char*)res + (size_t)Mfull * ldr * rbytes : nullptr;
            t.M = tail;
            t.tiles_m = 1; t.tiles_n = (N + BN - 1) / BN;
            hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)t.tiles_n), dim3(256), 0, (hipStream_t)stream, t);
            return (int)hipGetLastError();
        }
        p.tiles_m = (M + TM - 1) / TM; p.tiles_n = (N + TN - 1) / TN;
        const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
        if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_bf16: grid too large"); return FW_E_BADARG; }
        const int var = fw_get_option(FW_OPT_GEMM_VAR);
        hipStream_t st = (hipStream_t)stream;
        if (kern == 4 && K >= 4 * BK) {
            if (var & 2) hipLaunchKernelGGL(gemm_bf16_pp2_kernel<true>, dim3((unsigned)nwg), dim3(512), 0, st, p);      // TIMING build
            else hipLaunchKernelGGL(gemm_bf16_pp2_kernel<false>, dim3((unsigned)nwg), dim3(512), 0, st, p);
            return (int)hipGetLastError();
        }
        if (kern == 3 && K >= 4 * BK) {
            if (var == 1) hipLaunchKernelGGL(gemm_bf16_pp_kernel<1>, dim3((unsigned)nwg), dim3(512), 0, st, p);
            else hipLaunchKernelGGL(gemm_bf16_pp_kernel<0>, dim3((unsigned)nwg), dim3(512), 0, st, p);
            return (int)hipGetLastError();
        }
        if (kern == 2 && K >= 8 * HK) {
            switch (var) {
                case 1: hipLaunchKernelGGL(gemm_bf16_w4_kernel<1>, dim3((unsigned)nwg), dim3(256), 0, st, p); break;
                case 4: hipLaunchKernelGGL(gemm_bf16_w4_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, st, p); break;
                default: hipLaunchKernelGGL(gemm_bf16_w4_kernel<0>, dim3((unsigned)nwg), dim3(256), 0, st, p); break;
            }
            return (int)hipGetLastError();
        }
        if (kern == 1 && K >= 8 * HK) {
            switch (var) {
                case 1: hipLaunchKernelGGL(gemm_bf16_ring_kernel<1>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 2: hipLaunchKernelGGL(gemm_bf16_ring_kernel<2>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 3: hipLaunchKernelGGL(gemm_bf16_ring_kernel<3>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 4: hipLaunchKernelGGL(gemm_bf16_ring_kernel<4>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 5: hipLaunchKernelGGL(gemm_bf16_ring_kernel<5>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 8: hipLaunchKernelGGL(gemm_bf16_ring_kernel<8>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 9: hipLaunchKernelGGL(gemm_bf16_ring_kernel<9>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 16: hipLaunchKernelGGL(gemm_bf16_ring_kernel<16>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 32: hipLaunchKernelGGL(gemm_bf16_ring_kernel<32>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 48: hipLaunchKernelGGL(gemm_bf16_ring_kernel<48>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                case 64: hipLaunchKernelGGL(gemm_bf16_ring_kernel<64>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
                default: hipLaunchKernelGGL(gemm_bf16_ring_kernel<0>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            }
            return (int)hipGetLastError();
        }
        switch (var) {
            case 1: hipLaunchKernelGGL(gemm_bf16_256_kernel<1>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 2: hipLaunchKernelGGL(gemm_bf16_256_kernel<2>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 3: hipLaunchKernelGGL(gemm_bf16_256_kernel<3>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 4: hipLaunchKernelGGL(gemm_bf16_256_kernel<4>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 6: hipLaunchKernelGGL(gemm_bf16_256_kernel<6>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 7: hipLaunchKernelGGL(gemm_bf16_256_kernel<7>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 8: hipLaunchKernelGGL(gemm_bf16_256_kernel<8>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            case 12: hipLaunchKernelGGL(gemm_bf16_256_kernel<12>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
            default: hipLaunchKernelGGL(gemm_bf16_256_kernel<0>, dim3((unsigned)nwg), dim3(512), 0, st, p); break;
        }
        return (int)hipGetLastError();
    }
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_bf16: grid too large"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// Measurement hook (tools/gemm_timeline.py): the phase timestamps written by the TIMING build of the ping-pong kernel.
extern "C" int fw_debug_gemm_timestamps(unsigned long long* host_out, int n) {
    if (n <= 0 || n > 128) { fw_set_error("fw_debug_gemm_timestamps: n must be in 1..128"); return FW_E_BADARG; }
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_ts), sizeof(unsigned long long) * n);
}

extern "C" int fw_gemv_f32(const float* x, const float* W, int64_t ldw, const float* bias, float* out,
                           int N, int K, int act_in_silu, int act_out, void* stream) {
    if (N <= 0 || K <= 0) { fw_set_error("fw_gemv_f32: bad shape"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, W, ldw, bias, out, N, K, act_in_silu, act_out);
    return (int)hipGetLastError();
}

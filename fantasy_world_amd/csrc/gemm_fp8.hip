// e4m3 x e4m3 MFMA GEMM with the fp8_linear epilogue for gfx950:  C = epi((A[M,K] W[N,K]^T) * scale_a[m])
//
// The reference's fp8 linear (AutoWrappedLinear.fp8_linear, FantasyWorld/diffsynth_wan22/vram_management/layers.py:115-151) is
// torch._scaled_mm(xq, wq^T, scale_a (per row), 1, bias) on e4m3 operands with fp32 accumulation.  Here the product runs on the
// block-scaled instruction v_mfma_scale_f32_32x32x64_f8f6f4 with every E8M0 block scale = 2^0 (operand byte 0x7f), which is
// the ONLY fp8 MFMA that runs at the fp8 rate on gfx950 (MI355X_MICROARCH.md MFMA table: the non-scaled 32x32x16 / 16x16x32
// forms run at the bf16 rate); the reference's per-row scale_a stays a multiply in the epilogue, exactly where _scaled_mm applies it.
//
// Schedule: the 256x256 ping-pong of gemm_bf16_pp_kernel (csrc/gemm.hip) with the k-slab doubled to 128 elements -- LDS rows stay
// 128 B (the LDS-DMA ingest sweet spot measured by tools/probes/dma_probe.hip), a slab is still four 16 KiB units streamed by
// global_load_lds_dwordx4 with counted vmcnt waits, but it now feeds 2 k-steps of 64 (16 MFMAs of 64 cycles per wave and phase pair
// instead of 32 of 32 cycles): the same bytes per slab for twice the matrix time, so the DMA ingest that bounds the bf16 kernel has
// twice the slack.  Operand layout of the 32x32x64 instruction (probed on the device, tools/probes/mfma_scale_probe.hip):
// lane l supplies row (l & 31), k = 32 (l >> 5) .. +31 as 32 consecutive bytes = two 16-B LDS chunks; C/D uses the standard
// 32x32 map.  LDS image: row r at r*128, 16-B chunk c stored at c ^ ((r >> 1) & 7) (swizzle applied to the DMA SOURCE address and
// again on the read): conflict-free ds_read_b128.
#include "fw_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

constexpr int QM = 256, QN = 256, QK = 128;        // tile; QK in fp8 elements = bytes
constexpr int QSTAGE = (QM + QN) * QK;             // 64 KiB

struct QArgs {
    const uint8_t* A; int64_t lda;
    const uint8_t* W; int64_t ldw;
    const float* scale_a;
    void* C; int64_t ldc; int out_dtype;
    int M, N, K;
    const float* bias; int act; const float* g1; const float* g0;
    const void* res; int64_t ldr; int res_dtype;
    int tiles_m, tiles_n;
};

#define FWQ_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N> __device__ __forceinline__ void fwq_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define FWQ_MFMA(A, B, C) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)

__device__ __forceinline__ i32x8_t fwq_frag(const char* row_base, int off0, int off1) {
    const i32x4_t lo = *(const i32x4_t*)(row_base + off0);
    const i32x4_t hi = *(const i32x4_t*)(row_base + off1);
    return i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// scale_a -> bias -> activation -> per-column affine -> + residual -> store, through a per-wave 16 KiB LDS region so that global
// traffic is row-contiguous (a lane owns 4 fixed columns of one row per step).
__device__ __forceinline__ void epilogue_fp8(const QArgs& p, char* smem, f32x16_t (&acc)[4][2], int wave, int grp, int wn,
                                             int fi, int hi, int lane, int m0, int n0) {
    char* reg = smem + wave * 16384;
    const int rl = lane >> 4;              // row inside a 4-row read group
    const int c4 = (lane & 15) * 4;        // first of this lane's 4 columns
    const int gcol = n0 + wn * 64 + c4;
    const bool col_ok = gcol < p.N;        // N % 4 == 0 (launcher)
    f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f}, g14 = {1.f, 1.f, 1.f, 1.f}, g04 = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
        if (p.bias) bias4 = *(const f32x4_t*)(p.bias + gcol);
        if (p.g1) g14 = *(const f32x4_t*)(p.g1 + gcol);
        if (p.g0) g04 = *(const f32x4_t*)(p.g0 + gcol);
    }
    const int act = p.act;
    if (p.res_dtype == FW_DT_F32) {
        // fp32 residual stream (o-projection / ffn2 of a DiT block): the scheme of the bf16 kernel's epilogue_256 -- all 16 residual
        // loads of a 64-row pass in flight before the LDS transpose, and pass 1's loads issued from inside pass 0's store loop, ahead
        // of the stores.  With the mainloop twice as fast as in bf16, the epilogue weighs twice as much here.
        f32x4_t rv[16];
        float sa[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = m0 + grp * 128 + it * 4 + rl;
            rv[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            sa[it] = 0.f;
            if (row < p.M && col_ok) { rv[it] = *(const f32x4_t*)((const float*)p.res + (int64_t)row * p.ldr + gcol); sa[it] = p.scale_a[row]; }
        }
        auto pass = [&](auto q_tag) {
            constexpr int q = decltype(q_tag)::value;              // compile-time: acc[] must never be indexed dynamically
#pragma unroll
            for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row_l = rb2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        *(float*)(reg + row_l * 256 + (nb * 32 + fi) * 4) = acc[2 * q + rb2][nb][r];
                    }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row_l = it * 4 + rl;
                f32x4_t v = *(const f32x4_t*)(reg + row_l * 256 + c4 * 4);
                const int row = m0 + grp * 128 + q * 64 + row_l;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fw_affine(v[j], sa[it], bias4[j]);
                if (act != FW_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fw_apply_act(v[j], act);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fw_affine(v[j], g14[j], g04[j]);
                v += rv[it];
                if (q == 0) {
                    const int row1 = row + 64;
                    rv[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    sa[it] = 0.f;
                    if (row1 < p.M && col_ok) { rv[it] = *(const f32x4_t*)((const float*)p.res + (int64_t)row1 * p.ldr + gcol); sa[it] = p.scale_a[row1]; }
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
        pass(std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, 1>{});
        return;
    }
    // no residual (qkv, ffn0 + GELU, ...) or a bf16 one: the activation is a compile-time parameter of the body, selected once per
    // wave -- with the switch inside the element loop every value walked a chain of scalar compares and taken branches (the bf16
    // kernel's epilogue_256 has the measurement).  -1 = decided per element (activation AND bf16 residual: not used by the forward).
    auto body = [&](auto act_tag, auto res_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr int RES = decltype(res_tag)::value;
        auto pass = [&](auto q_tag) __attribute__((always_inline)) {
            constexpr int q = decltype(q_tag)::value;              // compile-time: acc[] must never be indexed dynamically
#pragma unroll
            for (int rb2 = 0; rb2 < 2; ++rb2)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row_l = rb2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        *(float*)(reg + row_l * 256 + (nb * 32 + fi) * 4) = acc[2 * q + rb2][nb][r];
                    }
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row_l = it * 4 + rl;
                f32x4_t v = *(const f32x4_t*)(reg + row_l * 256 + c4 * 4);
                const int row = m0 + grp * 128 + q * 64 + row_l;
                if (row < p.M && col_ok) {
                    const float s = p.scale_a[row];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fw_affine(v[j], s, bias4[j]);
                    if (ACT < 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fw_apply_act(v[j], act);
                    } else if (ACT == FW_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    } else if (ACT == FW_ACT_GELU_TANH) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fw_gelu_tanh(v[j]);
                    } else if (ACT == FW_ACT_GELU_ERF) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fw_gelu_erf(v[j]);
                    } else if (ACT == FW_ACT_SILU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fw_silu(v[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fw_affine(v[j], g14[j], g04[j]);
                    if (RES == FW_DT_BF16) {
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
        };
        pass(std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, 1>{});
    };
    using R0 = std::integral_constant<int, FW_DT_NONE>;
    if (p.res_dtype == FW_DT_BF16) {
        if (act == FW_ACT_NONE) body(std::integral_constant<int, FW_ACT_NONE>{}, std::integral_constant<int, FW_DT_BF16>{});
        else body(std::integral_constant<int, -1>{}, std::integral_constant<int, FW_DT_BF16>{});
    } else {
        switch (act) {
            case FW_ACT_RELU: body(std::integral_constant<int, FW_ACT_RELU>{}, R0{}); break;
            case FW_ACT_GELU_TANH: body(std::integral_constant<int, FW_ACT_GELU_TANH>{}, R0{}); break;
            case FW_ACT_GELU_ERF: body(std::integral_constant<int, FW_ACT_GELU_ERF>{}, R0{}); break;
            case FW_ACT_SILU: body(std::integral_constant<int, FW_ACT_SILU>{}, R0{}); break;
            default: body(std::integral_constant<int, FW_ACT_NONE>{}, R0{}); break;
        }
    }
}

// 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 accumulators of 32x32.  The two 4-wave groups run one barrier slot apart
// (LOAD || MFMA ping-pong), exactly the slot table of gemm_bf16_pp_kernel:
//     slot:        4t        4t+1      4t+2      4t+3      4t+4
//     group A:   LOAD0(t)  MFMA0(t)  LOAD1(t)  MFMA1(t)  LOAD0(t+1)      LOAD0 reads B(t) + A rows 0-63 of the wave tile,
//     group B:   MFMA1(..) LOAD0(t)  MFMA0(t)  LOAD1(t)  MFMA1(t)        LOAD1 reads A rows 64-127
// with the 6 pieces of B(t+2) / A0(t+2) issued from inside MFMA1(t), the 2 pieces of A1(t+2) from MFMA0(t+1), and the counted
// waits 8 / 6 (group A) and 2 / 6 (group B) ahead of the barrier that precedes the first read of a unit.
__global__ __launch_bounds__(512, 2) void gemm_fp8_pp_kernel(QArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * QSTAGE];

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
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;     // XCD-contiguous tile ranges (bijective)
    }
    int tm, tn;
    {
        const int GROUP = p.tiles_n >= 20 ? 4 : 8;      // the bf16 kernel's measurement (gemm.hip: 4 M-tiles per group for wide outputs)
        const int per_group = GROUP * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP;
        const int gsz = min(p.tiles_m - first_m, GROUP);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * QM, n0 = tn * QN;

    // DMA pieces (1 KiB = 8 rows x 128 B).  Wave w owns, in every 128-row unit, pieces 2w and 2w+1 (rows 16w .. 16w+15).
    const char* abase = (const char*)p.A + (int64_t)m0 * p.lda;
    const char* wbase = (const char*)p.W + (int64_t)n0 * p.ldw;
    unsigned aoff[2][2], woff[2][2];       // [unit][piece], bytes
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = u * 128 + (wave * 2 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            aoff[u][i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 16);
            woff[u][i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 16);
        }
#define FWQ_A(S, KT, U, I) FW_GLDS16(abase + (size_t)(KT) * QK + aoff[U][I], smem + (S) * QSTAGE + ((U) * 16 + wave * 2 + (I)) * 1024)
#define FWQ_W(S, KT, U, I) FW_GLDS16(wbase + (size_t)(KT) * QK + woff[U][I], smem + (S) * QSTAGE + QM * QK + ((U) * 16 + wave * 2 + (I)) * 1024)
#define FWQ_ISSUE6(S, KT) do { FWQ_W(S, KT, 0, 0); FWQ_W(S, KT, 0, 1); FWQ_W(S, KT, 1, 0); FWQ_W(S, KT, 1, 1); FWQ_A(S, KT, 0, 0); FWQ_A(S, KT, 0, 1); } while (0)
#define FWQ_ISSUE2(S, KT) do { FWQ_A(S, KT, 1, 0); FWQ_A(S, KT, 1, 1); } while (0)

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int c0[2], c1[2];                      // byte offsets of the two 16-B chunks of k-step s inside a row
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        c0[s] = ((4 * s + 2 * hi) ^ swz) << 4;
        c1[s] = ((4 * s + 2 * hi + 1) ^ swz) << 4;
    }
    const int a_row_off = (grp * 128 + fi) * 128;                    // + rb*32*128, rb = 0..3
    const int b_row_off = QM * QK + (wn * 64 + fi) * 128;            // + nb*32*128

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    i32x8_t afr[2][2], bfr[2][2];          // [row / col block][k-step]

    const int nk = p.K / QK;               // >= 4 (launcher)
    FWQ_ISSUE6(0, 0); FWQ_ISSUE2(0, 0);
    FWQ_ISSUE6(1, 1); FWQ_ISSUE2(1, 1);
    fwq_wait_vm<8>();
    FWQ_BARRIER();
    if (grp == 1) FWQ_BARRIER();

    // One slab.  The three conditions are compile-time (the loop is peeled by hand: first / steady / second-last / last slab), so
    // a phase is ONE basic block: with run-time branches inside, LLVM sinks every MFMA of the slab past the barriers into the last
    // block and the ping-pong degenerates (seen in the ISA).  After each burst the accumulators are pinned by an empty asm, which
    // keeps the MFMAs on their side of the barrier that follows.
    int kt = 0;
    auto slab = [&](auto first_tag, auto has1_tag, auto has2_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, HAS1 = decltype(has1_tag)::value, HAS2 = decltype(has2_tag)::value;
        const int st = kt & 1;
        const char* base = smem + st * QSTAGE;
        // ---------------- LOAD0(kt): B fragments + A rows 0..63 of the wave tile
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bfr[0][s] = fwq_frag(base + b_row_off, c0[s], c1[s]);
            bfr[1][s] = fwq_frag(base + b_row_off + 32 * 128, c0[s], c1[s]);
            afr[0][s] = fwq_frag(base + a_row_off, c0[s], c1[s]);
            afr[1][s] = fwq_frag(base + a_row_off + 32 * 128, c0[s], c1[s]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 0) { if (HAS1) fwq_wait_vm<6>(); else fwq_wait_vm<0>(); }
        FWQ_BARRIER();
        // ---------------- MFMA0(kt) (+ A1 unit of slab kt+1 into the other stage; slabs 0 and 1 come from the prologue)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[0][0] = FWQ_MFMA(afr[0][s], bfr[0][s], acc[0][0]);
            acc[0][1] = FWQ_MFMA(afr[0][s], bfr[1][s], acc[0][1]);
            if (s == 0 && !FIRST && HAS1) FWQ_ISSUE2(st ^ 1, kt + 1);
            acc[1][0] = FWQ_MFMA(afr[1][s], bfr[0][s], acc[1][0]);
            acc[1][1] = FWQ_MFMA(afr[1][s], bfr[1][s], acc[1][1]);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        FWQ_BARRIER();
        // ---------------- LOAD1(kt): A rows 64..127 of the wave tile
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            afr[0][s] = fwq_frag(base + a_row_off + 64 * 128, c0[s], c1[s]);
            afr[1][s] = fwq_frag(base + a_row_off + 96 * 128, c0[s], c1[s]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) { if (HAS1) fwq_wait_vm<2>(); else fwq_wait_vm<0>(); }
        FWQ_BARRIER();
        // ---------------- MFMA1(kt) (+ B and A0 units of slab kt+2 into this stage)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[2][0] = FWQ_MFMA(afr[0][s], bfr[0][s], acc[2][0]);
            acc[2][1] = FWQ_MFMA(afr[0][s], bfr[1][s], acc[2][1]);
            if (s == 0 && HAS2) FWQ_ISSUE6(st, kt + 2);
            acc[3][0] = FWQ_MFMA(afr[1][s], bfr[0][s], acc[3][0]);
            acc[3][1] = FWQ_MFMA(afr[1][s], bfr[1][s], acc[3][1]);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        if (grp == 0) { if (HAS2) fwq_wait_vm<8>(); else if (HAS1) fwq_wait_vm<2>(); else fwq_wait_vm<0>(); }
        else { if (HAS2) fwq_wait_vm<6>(); else fwq_wait_vm<0>(); }
        FWQ_BARRIER();
        ++kt;
    };
    using T = std::true_type;
    using F = std::false_type;
    slab(T{}, T{}, T{});                                   // kt = 0 (nk >= 4)
    while (kt < nk - 2) slab(F{}, T{}, T{});
    slab(F{}, T{}, F{});                                   // kt = nk - 2
    slab(F{}, F{}, F{});                                   // kt = nk - 1
    if (grp == 0) FWQ_BARRIER();
    epilogue_fp8(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_fp8_two_slot_kernel (round 6, the default): the TWO-SLOT schedule of gemm_bf16_two_slot_kernel (csrc/gemm_pp.hip) on the fp8
// operands.  The four-slot kernel above keeps the matrix pipe 61 % busy at 1.92 GHz (profiles/r05/clock_probe_call21_with_fp8_kernels.txt);
// the same transformation took the bf16 GEMM from 74 % to 88 %.  The LDS image is byte-identical to the bf16 kernel's (a 128-B row is 64
// bf16 there and 128 e4m3 here; A in two 32 KiB stages, W in THREE), and so is the slot table:
//     slot:        2t          2t+1         2t+2         2t+3
//     group A:   LOAD(t)     MFMA(t)      LOAD(t+1)    MFMA(t+1)          LOAD(t): W(t) fragments + A rows 0..63 of the wave tile, then
//     group B:   MFMA(t-1)   LOAD(t)      MFMA(t)      LOAD(t+1)          this wave's 8 LDS-DMA pieces; MFMA(t): 16 MFMAs of 64 cycles in
//                                                                         ONE burst, A rows 64..127 re-read into the consumed registers
//   * two barriers per k-slab instead of four; no LDS-DMA instruction inside a burst (group A, LOAD(t): A0(t+1) + W rows 0..127 of
//     slab t+2; group B, LOAD(t): A1(t+1) + W rows 128..255 of slab t+2 -- the third W stage is what lets W(t+3) be requested as soon
//     as W(t) has been read);
//   * pieces by buffer_load ... lds through an SGPR descriptor (rows past M / N are out of range: zeros, no clamps);
//   * counted waits: group A vmcnt(4) at the end of its burst, group B vmcnt(8) at the end of LOAD and vmcnt(4) at the end of its burst.
// Same k order per output element as the four-slot kernel (k-step 0 then 1 of every slab into the same accumulator): BIT-identical
// results (tests/test_fp8_gpu.py).  FW_GEMM_KERNEL=4 selects the four-slot kernel for A/B.
// ---------------------------------------------------------------------------------------------------------------
#define FWQ_BLDS16(rs, voff, soff, ldsptr) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, FW_LDS_PTR(ldsptr), 16, voff, soff, 0, 0)
constexpr int Q2_A_STAGE = QM * QK;                    // 32 KiB
constexpr int Q2_W_BASE = 2 * Q2_A_STAGE;              // 64 KiB
constexpr int Q2_W_STAGE = QN * QK;                    // 32 KiB
constexpr int Q2_LDS = Q2_W_BASE + 3 * Q2_W_STAGE;     // 160 KiB

__global__ __launch_bounds__(512, 2) void gemm_fp8_two_slot_kernel(QArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[Q2_LDS];

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
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;     // XCD-contiguous tile ranges (bijective)
    }
    int tm, tn;
    {
        const int GROUP = p.tiles_n >= 20 ? 4 : 8;
        const int per_group = GROUP * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GROUP;
        const int gsz = min(p.tiles_m - first_m, GROUP);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * QM, n0 = tn * QN;

    const long long arem = (long long)(p.M - m0) * p.lda, wrem = (long long)(p.N - n0) * p.ldw;            // bytes
    const auto ars = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0,
                                                       (int)(unsigned)(arem > 0xffffffffLL ? 0xffffffffLL : arem), 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw), 0,
                                                       (int)(unsigned)(wrem > 0xffffffffLL ? 0xffffffffLL : wrem), 0x00020000);
    // a piece = 8 rows x 128 B; lane -> (row pr, 16-B chunk pc); source-side swizzle chunk ^ ((row >> 1) & 7) for the even / odd piece
    const int pr = lane >> 3, pc = lane & 7;
    const int va[2] = {pr * (int)p.lda + ((pc ^ (pr >> 1)) << 4), pr * (int)p.lda + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int vw[2] = {pr * (int)p.ldw + ((pc ^ (pr >> 1)) << 4), pr * (int)p.ldw + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int astep = __builtin_amdgcn_readfirstlane((int)p.lda * 8), wstep = __builtin_amdgcn_readfirstlane((int)p.ldw * 8);
    // this wave's 4 A pieces and 4 W pieces of every slab: tile rows (grp * 128 + 32 wn) .. + 31 of both operands
    const int q0 = grp * 16 + 4 * wn;
#define FWQ2_A(AS, KT, J) FWQ_BLDS16(ars, va[(J) & 1], (KT) * QK + (q0 + (J)) * astep, smem + (AS) * Q2_A_STAGE + (q0 + (J)) * 1024)
#define FWQ2_W(WS, KT, J) FWQ_BLDS16(wrs, vw[(J) & 1], (KT) * QK + (q0 + (J)) * wstep, smem + Q2_W_BASE + (WS) * Q2_W_STAGE + (q0 + (J)) * 1024)
    auto issue_a = [&](int as, int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) FWQ2_A(as, kt, j);
    };
    auto issue_w = [&](int ws, int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) FWQ2_W(ws, kt, j);
    };

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int c0[2], c1[2];                      // byte offsets of the two 16-B chunks of k-step s inside a row
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        c0[s] = ((4 * s + 2 * hi) ^ swz) << 4;
        c1[s] = ((4 * s + 2 * hi + 1) ^ swz) << 4;
    }
    const int a_row_off = (grp * 128 + fi) * 128;
    const int b_row_off = Q2_W_BASE + (wn * 64 + fi) * 128;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    i32x8_t afr[2][2], bfr[2][2];          // [row / col block][k-step]

    const int nk = p.K / QK;               // >= 4 (launcher)

    // ---- prologue: A(0), W(0), A(1), W(1), W(2): every wave its own pieces, in the order the steady-state waits assume ----
    issue_a(0, 0); issue_w(0, 0);
    issue_a(1, 1); issue_w(1, 1);
    issue_w(2, 2);
    fwq_wait_vm<12>();
    FWQ_BARRIER();
    if (grp == 1) FWQ_BARRIER();

    int kt = 0;
    int ws = 0;                            // W stage of slab kt = kt % 3
    auto slab = [&](auto g_tag, auto first_tag, auto n1_tag, auto n2_tag) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, N1 = decltype(n1_tag)::value, N2 = decltype(n2_tag)::value;   // kt == 0; kt + 1 < nk; kt + 2 < nk
        const int as = kt & 1;
        const char* abase = smem + as * Q2_A_STAGE;
        const char* bbase = smem + ws * Q2_W_STAGE;
        // ---------------- LOAD(kt): W fragments + A rows 0..63 of the wave tile, then this wave's 8 pieces
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bfr[0][s] = fwq_frag(bbase + b_row_off, c0[s], c1[s]);
            bfr[1][s] = fwq_frag(bbase + b_row_off + 32 * 128, c0[s], c1[s]);
            afr[0][s] = fwq_frag(abase + a_row_off, c0[s], c1[s]);
            afr[1][s] = fwq_frag(abase + a_row_off + 32 * 128, c0[s], c1[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!FIRST && N1) {
            issue_a(as ^ 1, kt + 1);                          // A0 / A1 of slab kt+1 -> the A stage slab kt-1 has left
            if (N2) issue_w(ws == 0 ? 2 : ws - 1, kt + 2);    // W half of slab kt+2 -> W stage (kt+2) % 3 = (kt-1) % 3
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (G == 1 && N1) {                                   // own W-hi(kt+1) landed (group A reads it in the next slot)
            if (FIRST) fwq_wait_vm<4>(); else if (N2) fwq_wait_vm<8>(); else fwq_wait_vm<4>();
        }
        FWQ_BARRIER();
        // ---------------- MFMA(kt): 16 MFMAs in one burst; A rows 64..127 re-read into the registers the first eight have consumed
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[0][0] = FWQ_MFMA(afr[0][s], bfr[0][s], acc[0][0]);
            acc[0][1] = FWQ_MFMA(afr[0][s], bfr[1][s], acc[0][1]);
            __builtin_amdgcn_sched_barrier(0);
            afr[0][s] = fwq_frag(abase + a_row_off + 64 * 128, c0[s], c1[s]);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = FWQ_MFMA(afr[1][s], bfr[0][s], acc[1][0]);
            acc[1][1] = FWQ_MFMA(afr[1][s], bfr[1][s], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            afr[1][s] = fwq_frag(abase + a_row_off + 96 * 128, c0[s], c1[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[2][0] = FWQ_MFMA(afr[0][s], bfr[0][s], acc[2][0]);
            acc[2][1] = FWQ_MFMA(afr[0][s], bfr[1][s], acc[2][1]);
            acc[3][0] = FWQ_MFMA(afr[1][s], bfr[0][s], acc[3][0]);
            acc[3][1] = FWQ_MFMA(afr[1][s], bfr[1][s], acc[3][1]);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        asm volatile("" : "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        if (N1) {                                             // own A pieces (and group A: W-lo) of slab kt+1 landed; W(kt+2) stays in flight
            if (N2) fwq_wait_vm<4>(); else fwq_wait_vm<0>();
        }
        FWQ_BARRIER();
        ++kt;
        ws = ws == 2 ? 0 : ws + 1;
    };
    using T = std::true_type;
    using F = std::false_type;
    auto mainloop = [&](auto g_tag) __attribute__((always_inline)) {
        // (distinct opaque markers at both ends of an arm: without them the compiler hoists the common first LOAD above the group
        //  branch and merges the arms' tails -- gemm_pp.hip has the finding)
        asm volatile("; fp8 two-slot mainloop, group %0: begin" ::"n"(decltype(g_tag)::value) : "memory");
        slab(g_tag, T{}, T{}, T{});
        while (kt < nk - 2) slab(g_tag, F{}, T{}, T{});
        slab(g_tag, F{}, T{}, F{});                           // kt = nk - 2
        slab(g_tag, F{}, F{}, F{});                           // kt = nk - 1
        asm volatile("; fp8 two-slot mainloop, group %0: end" ::"n"(decltype(g_tag)::value) : "memory");
    };
    if (grp == 0) { mainloop(std::integral_constant<int, 0>{}); FWQ_BARRIER(); }
    else mainloop(std::integral_constant<int, 1>{});
    epilogue_fp8(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// Small / ragged shapes: 128x128x64 tile, 4 waves, register prefetch, padded LDS rows, v_mfma_f32_32x32x16_fp8_fp8 (bf16 rate;
// these GEMMs are launch / HBM bound).  Same epilogue semantics.
// ---------------------------------------------------------------------------------------------------------------
constexpr int FBM = 128, FBN = 128, FBK = 64;
constexpr int FROW = 72;                          // padded LDS row (bytes): 18 dwords -> conflict-free 8-byte fragment reads

__global__ __launch_bounds__(256) void gemm_fp8_small_kernel(QArgs p) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][2][FBM * FROW];      // [stage][A | W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
    const int m0 = tm * FBM, n0 = tn * FBN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fi = lane & 31, hi = lane >> 5;

    const int lrow = tid >> 2, lchunk = (tid & 3) * 16;
    const uint8_t* ga[2];
    const uint8_t* gw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ga[i] = p.A + (int64_t)min(m0 + lrow + 64 * i, p.M - 1) * p.lda + lchunk;
        gw[i] = p.W + (int64_t)min(n0 + lrow + 64 * i, p.N - 1) * p.ldw + lchunk;
    }
    u32x4_t ra[2], rw[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = *(const u32x4_t*)(ga[i] + k0);
            rw[i] = *(const u32x4_t*)(gw[i] + k0);
        }
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint8_t* da = &lds[st][0][(lrow + 64 * i) * FROW + lchunk];
            uint8_t* dw = &lds[st][1][(lrow + 64 * i) * FROW + lchunk];
            *(u32x2_t*)da = u32x2_t{ra[i][0], ra[i][1]};
            *(u32x2_t*)(da + 8) = u32x2_t{ra[i][2], ra[i][3]};
            *(u32x2_t*)dw = u32x2_t{rw[i][0], rw[i][1]};
            *(u32x2_t*)(dw + 8) = u32x2_t{rw[i][2], rw[i][3]};
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = p.K / FBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * FBK);
#pragma unroll
        for (int s = 0; s < FBK / 16; ++s) {
            long fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const long*)&lds[st][0][(wm + 32 * a + fi) * FROW + s * 16 + hi * 8];
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = *(const long*)&lds[st][1][(wn + 32 * b + fi) * FROW + s * 16 + hi * 8];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            lstore(st ^ 1);
            __syncthreads();
        }
    }

    // D[m][n] with n = lane % 32 (column of the W-side operand), m = (r % 4) + 8 (r / 4) + 4 (lane / 32)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn + 32 * b + fi;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
            const float g1 = p.g1 ? p.g1[n] : 1.f;
            const float g0 = p.g0 ? p.g0[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= p.M) continue;
                float y = fw_affine(acc[a][b][r], p.scale_a[m], bv);
                y = fw_apply_act(y, p.act);
                y = fw_affine(y, g1, g0);
                if (p.res_dtype == FW_DT_F32) y += ((const float*)p.res)[(int64_t)m * p.ldr + n];
                else if (p.res_dtype == FW_DT_BF16) y += bf16_bits_to_f32(((const uint16_t*)p.res)[(int64_t)m * p.ldr + n]);
                if (p.out_dtype == FW_DT_F32) ((float*)p.C)[(int64_t)m * p.ldc + n] = y;
                else ((uint16_t*)p.C)[(int64_t)m * p.ldc + n] = f32_to_bf16_bits(y);
            }
        }
}

}  // namespace

extern "C" int fw_gemm_fp8(const uint8_t* A, int64_t lda, const uint8_t* W, int64_t ldw, const float* scale_a,
                           void* C, int64_t ldc, int out_dtype, int M, int N, int K,
                           const float* bias, int act, const float* g1, const float* g0,
                           const void* res, int64_t ldr, int res_dtype, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A || !W || !scale_a || !C || K <= 0 || (K % FBK) || (lda % 16) || (ldw % 16) || (((uintptr_t)A) & 15) || (((uintptr_t)W) & 15)) {
        fw_set_error("fw_gemm_fp8: K must be a positive multiple of 64; A/W 16-byte aligned with lda/ldw % 16 == 0"); return FW_E_BADARG; }
    if (out_dtype != FW_DT_BF16 && out_dtype != FW_DT_F32) { fw_set_error("fw_gemm_fp8: bad out_dtype"); return FW_E_BADARG; }
    if (res_dtype != FW_DT_NONE && res == nullptr) { fw_set_error("fw_gemm_fp8: res_dtype set but res NULL"); return FW_E_BADARG; }
    QArgs p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.scale_a = scale_a; p.C = C; p.ldc = ldc; p.out_dtype = out_dtype;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.act = act; p.g1 = g1; p.g0 = g0;
    p.res = res; p.ldr = ldr; p.res_dtype = res ? res_dtype : FW_DT_NONE;
    hipStream_t st = (hipStream_t)stream;
    bool big = M >= 2048 && N >= 1024 && (K % QK) == 0 && K >= 4 * QK && fw_get_option(FW_OPT_GEMM_TILE) != 128;
    const uintptr_t cmask = (out_dtype == FW_DT_F32) ? 15 : 7;
    const uintptr_t rmask = (res_dtype == FW_DT_F32) ? 15 : 7;
    if ((N % 4) || (ldc % 4) || (((uintptr_t)C) & cmask) || (res && ((ldr % 4) || (((uintptr_t)res) & rmask)))) big = false;
    if ((((uintptr_t)bias) | ((uintptr_t)g1) | ((uintptr_t)g0)) & 15) big = false;
    int m_big = 0;
    if (big) {
        // short M tail (<= 128 rows past the last full 256-row band): peeled off to the small kernel instead of a whole extra
        // round of 256x256 tiles (same rule as fw_gemm_bf16)
        const int tail = M % QM;
        m_big = (tail > 0 && tail <= FBM && M >= 2 * QM) ? M - tail : M;
        QArgs b = p;
        b.M = m_big;
        b.tiles_m = (m_big + QM - 1) / QM; b.tiles_n = (N + QN - 1) / QN;
        const int64_t nwg = (int64_t)b.tiles_m * b.tiles_n;
        if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_fp8: grid too large"); return FW_E_BADARG; }
        // default: the two-slot kernel (32-bit byte offsets inside a tile: lda, ldw < 2^23); FW_GEMM_KERNEL=4: the four-slot kernel (A/B)
        if (fw_get_option(FW_OPT_GEMM_KERNEL) != 4 && lda < (1 << 23) && ldw < (1 << 23))
            hipLaunchKernelGGL(gemm_fp8_two_slot_kernel, dim3((unsigned)nwg), dim3(512), 0, st, b);
        else
            hipLaunchKernelGGL(gemm_fp8_pp_kernel, dim3((unsigned)nwg), dim3(512), 0, st, b);
        int rc = (int)hipGetLastError();
        if (rc || m_big == M) return rc;
    }
    QArgs t = p;
    const size_t cbytes = (out_dtype == FW_DT_F32) ? 4 : 2, rbytes = (res_dtype == FW_DT_F32) ? 4 : 2;
    t.A = A + (int64_t)m_big * lda;
    t.scale_a = scale_a + m_big;
    t.C = (char*)C + (size_t)m_big * ldc * cbytes;
    t.res = res ? (const char*)res + (size_t)m_big * ldr * rbytes : nullptr;
    t.M = M - m_big;
    t.tiles_m = (t.M + FBM - 1) / FBM; t.tiles_n = (N + FBN - 1) / FBN;
    const int64_t nwg = (int64_t)t.tiles_m * t.tiles_n;
    if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_fp8: grid too large"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemm_fp8_small_kernel, dim3((unsigned)nwg), dim3(256), 0, st, t);
    return (int)hipGetLastError();
}

// bf16 MFMA GEMM with fused epilogue for gfx950:  C = epi(A[M,K] * W[N,K]^T)
//
// Three kernels (round 1 carried six; the superseded generations -- first 2-stage staggered kernel, 5-deep half-slab ring,
// compiler-ordered four-wave kernel, first ping-pong kernel -- were removed in round 2, their numbers are in docs/history/DESIGN_r03.md section 4):
//   gemm_bf16_four_slot_kernel   256x256x64, 8 waves in two groups running LOAD || MFMA ping-pong: every big token-major GEMM (default)
//   gemm_bf16_w4b_kernel   256x256x64, 4 waves with 128x128 wave tiles, hand-ordered single instruction stream: the INDEPENDENT
//                          implementation the full-size agreement tests compare the default path with (FW_GEMM_KERNEL=5)
//   gemm_bf16_kernel       128x128x64, 4 waves: small / ragged shapes and the <= 128-row M tail of the big ones
// All stream A and W k-slabs HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip).  LDS rows are 128 B (64 bf16); the
// 16-byte chunk index is XOR-swizzled with ((row>>1)&7) so that the ds_read_b128 fragment reads of a 16-lane service group hit 16
// distinct 16-B slots of the 256-B bank row (conflict free).  Because global_load_lds writes lane-linear, the swizzle is applied
// to the per-lane SOURCE address and again on the read (both sides or neither).  Work-group ids are remapped so each XCD
// (private L2) owns a contiguous range of tiles, grouped 8 M-tiles deep so concurrently resident tiles share A bands and W panels.
#include "gemm_common.h"

using namespace fwgemm;

namespace {

__device__ __attribute__((aligned(128))) uint16_t g_conv_zero[64];     // the 128-B row every out-of-volume tap reads

// (absolute frame | yo * sh | xo * sw) of output row r, 8 + 12 + 12 bits (the launcher checks the ranges)
__device__ __forceinline__ unsigned conv_row_pack(const ConvGeom& g, int r) {
    const int hw = g.Ho * g.Wo;
    const int tt = r / hw, rem = r - tt * hw;
    const int yo = rem / g.Wo, xo = rem - yo * g.Wo;
    return ((unsigned)(g.t0 + tt) << 24) | ((unsigned)(yo * g.sh) << 12) | (unsigned)(xo * g.sw);
}

// Position of a k-slab in the tap walk (wave-uniform: lives in SGPRs); slabs are visited in order, so no division is needed.
struct ConvTap { int cc, dx, dy, dt; };
__device__ __forceinline__ void conv_tap_next(const ConvGeom& g, ConvTap& t) {
    // selects, not branches: this runs inside the MFMA bursts of the ping-pong kernel, whose phases must stay single basic blocks
    const int cc = t.cc + 1;
    const int wc = cc == g.cpt;
    t.cc = wc ? 0 : cc;
    const int dx = t.dx + wc;
    const int wx = dx == g.kw;
    t.dx = wx ? 0 : dx;
    const int dy = t.dy + wx;
    const int wy = dy == g.kh;
    t.dy = wy ? 0 : dy;
    t.dt += wy;
}

// Source of this lane's 16 bytes of a DMA piece: row `pack` of the output, slab position `t`, byte offset of the (swizzled) chunk
__device__ __forceinline__ const char* conv_src(const GemmArgs& p, unsigned pack, const ConvTap& t, int chunk_bytes) {
    const ConvGeom& g = p.cv;
    const int ti = (int)(pack >> 24) + t.dt - (g.kt - 1);
    const int yi = (int)((pack >> 12) & 0xfffu) + t.dy - g.ph;
    const int xi = (int)(pack & 0xfffu) + t.dx - g.pw;
    const bool ok = (unsigned)ti < (unsigned)g.T && (unsigned)yi < (unsigned)(g.H << g.ups) && (unsigned)xi < (unsigned)(g.W << g.ups);
    const int64_t row = ((int64_t)ti * g.H + (yi >> g.ups)) * g.W + (xi >> g.ups);
    const char* src = (const char*)p.A + (row * p.lda + t.cc * 64) * 2 + chunk_bytes;
    return ok ? src : (const char*)g_conv_zero + chunk_bytes;
}

// NST = 2: two 32 KiB stages, two work-groups per CU -- grids of many tiles.  NST = 4: a 4-deep ring with counted waits (three slabs
// in flight) and the CU to itself -- the launches with at most one round of tiles (the <= 128-row M tails of the VGGT / bicross
// GEMMs, context K/V, embeddings), where a work-group walks K alone and the 2-stage loop pays a DMA round trip per slab.  Same k
// order per output element either way (bit-identical results).
template <bool CONV, int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE_BYTES];

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
    unsigned apack[4];                     // CONV: the output pixel of each of this lane's 4 tile rows
    int achunk[4];
    ConvTap tap = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave * 4 + i;
        const int row = pc * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int ar = min(m0 + row, p.M - 1);
        const int wr = min(n0 + row, p.N - 1);
        if (CONV) { apack[i] = conv_row_pack(p.cv, ar); achunk[i] = chunk * 16; }
        else ag[i] = p.A + (int64_t)ar * p.lda + chunk * 8;
        wg_[i] = p.W + (int64_t)wr * p.ldw + chunk * 8;
    }

    auto stage = [&](int s) {
        char* a_lds = smem + s * STAGE_BYTES;
        char* b_lds = a_lds + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave * 4 + i;
            if (CONV) {
                FW_GLDS16(conv_src(p, apack[i], tap, achunk[i]), a_lds + pc * 1024);
            } else {
                FW_GLDS16(ag[i], a_lds + pc * 1024);
                ag[i] += BK;
            }
            FW_GLDS16(wg_[i], b_lds + pc * 1024);
            wg_[i] += BK;
        }
        if (CONV) conv_tap_next(p.cv, tap);
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
    if (NST == 2) {
        stage(0);
        __syncthreads();
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NST - 1; ++s0)
            if (s0 < nk) stage(s0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        if (NST == 2) {
            if (kt + 1 < nk) stage((kt + 1) & 1);
        } else {
            // slab kt has landed once at most the NST - 2 slabs requested after it are still in flight (8 DMA pieces per wave and slab)
            if (kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // visible to all; everyone is done reading slab kt - 1
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NST - 1 < nk) stage((kt + NST - 1) % NST);
        }
        const char* base = smem + (kt % NST) * STAGE_BYTES;
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
        if (NST == 2) __syncthreads();   // next slab landed (vmcnt(0)) and every wave is done reading this one
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------
    // (the activation is a compile-time parameter of the body, selected once: see epilogue_256)
    auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
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
                        v = fw_apply_act_ct<ACT>(v);
                        v = fw_affine(v, g1, g0);
                        if (p.res_dtype == FW_DT_F32) v += ((const float*)p.res)[(int64_t)row * p.ldr + col];
                        else if (p.res_dtype == FW_DT_BF16) v += bf16_bits_to_f32(((const uint16_t*)p.res)[(int64_t)row * p.ldr + col]);
                        if (p.out_dtype == FW_DT_F32) ((float*)p.C)[(int64_t)row * p.ldc + col] = v;
                        else ((uint16_t*)p.C)[(int64_t)row * p.ldc + col] = f32_to_bf16_bits(v);
                    }
                }
            }
        }
    };
    switch (p.act) {
        case FW_ACT_RELU: epilogue(std::integral_constant<int, FW_ACT_RELU>{}); break;
        case FW_ACT_GELU_TANH: epilogue(std::integral_constant<int, FW_ACT_GELU_TANH>{}); break;
        case FW_ACT_GELU_ERF: epilogue(std::integral_constant<int, FW_ACT_GELU_ERF>{}); break;
        case FW_ACT_SILU: epilogue(std::integral_constant<int, FW_ACT_SILU>{}); break;
        default: epilogue(std::integral_constant<int, FW_ACT_NONE>{}); break;
    }
}


// Epilogue of the four-wave kernel (wave tile 128 x 128): four 32-row passes through a private 16 KiB LDS region.
__device__ __forceinline__ void epilogue_w4(const GemmArgs& p, char* smem, f32x16_t (&acc)[4][4], int wave, int wm, int wn,
                                            int fi, int hi, int lane, int m0, int n0) {
    // Pass rb: the wave's 32 x 128 fp32 block goes through a private 16 KiB LDS region (raw accumulators in, row-contiguous
    // 16 B per lane out); bias / activation / per-column affine / residual are applied on the way out, where a lane owns 4
    // fixed columns and whole 512-B (fp32) / 256-B (bf16) row segments are read and written.  All 16 residual loads of a pass are
    // issued BEFORE the LDS transpose (the fp32 stream's HBM latency is paid once per pass), and the next pass's loads before this
    // pass's stores.
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
    auto run = [&](auto res_tag, auto act_tag) __attribute__((always_inline)) {
        constexpr int RES = decltype(res_tag)::value;          // FW_DT_NONE / FW_DT_BF16 / FW_DT_F32
        constexpr int ACT = decltype(act_tag)::value;          // FW_ACT_*, or -1 = decided per element at run time (rare combinations)
        // residual values of TWO passes at a time (32 loads = 32 KiB per wave, 128 KiB per CU in flight: with four waves per CU one
        // pass alone leaves the HBM latency half exposed; the fragment / staging registers of the mainloop are dead here)
        f32x4_t rv[2][16];
        u32x2_t rw[2][16];
        auto load_res = [&](auto rb_tag) __attribute__((always_inline)) {
            constexpr int rb = decltype(rb_tag)::value;
            if (RES == FW_DT_NONE) return;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = m0 + wm * 128 + rb * 32 + it * 2 + rl;
                const bool ok = row < p.M && col_ok;
                if (RES == FW_DT_F32) {
                    rv[rb & 1][it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (ok) rv[rb & 1][it] = *(const f32x4_t*)((const float*)p.res + (int64_t)row * p.ldr + gcol);
                } else {
                    rw[rb & 1][it] = u32x2_t{0u, 0u};
                    if (ok) rw[rb & 1][it] = *(const u32x2_t*)((const uint16_t*)p.res + (int64_t)row * p.ldr + gcol);
                }
            }
        };
        auto pass = [&](auto rb_tag) __attribute__((always_inline)) {
            constexpr int rb = decltype(rb_tag)::value;          // compile-time: acc[] must never be indexed dynamically
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row_l = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    *(float*)(reg + row_l * 512 + (nb * 32 + fi) * 4) = acc[rb][nb][r];
                }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row_l = it * 2 + rl;
                f32x4_t v = *(const f32x4_t*)(reg + row_l * 512 + c4 * 4);
                const int row = m0 + wm * 128 + rb * 32 + row_l;
                v += bias4;
                if (ACT < 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fw_apply_act(v[j], act);
                } else if (ACT != FW_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fw_apply_act_ct<(ACT < 0 ? 0 : ACT)>(v[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fw_affine(v[j], g14[j], g04[j]);
                if (RES == FW_DT_F32) {
                    v += rv[rb & 1][it];
                } else if (RES == FW_DT_BF16) {
                    v[0] += __uint_as_float(rw[rb & 1][it][0] << 16); v[1] += __uint_as_float(rw[rb & 1][it][0] & 0xffff0000u);
                    v[2] += __uint_as_float(rw[rb & 1][it][1] << 16); v[3] += __uint_as_float(rw[rb & 1][it][1] & 0xffff0000u);
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
        using R0 = std::integral_constant<int, 0>;
        using R1 = std::integral_constant<int, 1>;
        using R2 = std::integral_constant<int, 2>;
        using R3 = std::integral_constant<int, 3>;
        load_res(R0{}); load_res(R1{});
        pass(R0{}); pass(R1{});
        load_res(R2{}); load_res(R3{});
        pass(R2{}); pass(R3{});
    };
    // the activation is a compile-time parameter of the body wherever the forward uses one (plain outputs); see epilogue_256
    using ActNone = std::integral_constant<int, FW_ACT_NONE>;
    using ActAny = std::integral_constant<int, -1>;
    if (p.res_dtype == FW_DT_F32) {
        if (act == FW_ACT_NONE) run(std::integral_constant<int, FW_DT_F32>{}, ActNone{}); else run(std::integral_constant<int, FW_DT_F32>{}, ActAny{});
    } else if (p.res_dtype == FW_DT_BF16) {
        if (act == FW_ACT_NONE) run(std::integral_constant<int, FW_DT_BF16>{}, ActNone{}); else run(std::integral_constant<int, FW_DT_BF16>{}, ActAny{});
    } else {
        using R0 = std::integral_constant<int, FW_DT_NONE>;
        switch (act) {
            case FW_ACT_RELU: run(R0{}, std::integral_constant<int, FW_ACT_RELU>{}); break;
            case FW_ACT_GELU_TANH: run(R0{}, std::integral_constant<int, FW_ACT_GELU_TANH>{}); break;
            case FW_ACT_GELU_ERF: run(R0{}, std::integral_constant<int, FW_ACT_GELU_ERF>{}); break;
            case FW_ACT_SILU: run(R0{}, std::integral_constant<int, FW_ACT_SILU>{}); break;
            default: run(R0{}, ActNone{}); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256x64 ping-pong kernel, 128-B LDS rows, two 64 KiB stages, quarter-slab DMA scheduling with counted waits.
//
// Measured on MI355X (tools/probes/dma_probe.hip): the LDS-DMA ingest of a CU tops out at ~107 GB/s with 128-B global
// rows but only ~67 GB/s with 64-B rows (57 vs 85 GB/s beside ds_read traffic), and the GEMM's cost scales with the BYTES
// it streams, not with the number of DMA instructions.  So this kernel keeps full 128-B rows (k-slab 64)
// and gets its prefetch distance from scheduling instead of from more LDS: a slab is four 16 KiB
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
// (CONV: the A source of slab KT comes from the tap walk -- `tap6` / `tap2` are the positions of the next slab the 6-piece and the
//  2-piece issue streams will request; both streams visit the slabs in order, each advances its own position after issuing)
#define FW_PP_A(S, KT, U, I, TAP) FW_GLDS16(CONV ? conv_src(p, aoff[U][I], TAP, achunk[U][I]) : abase + (size_t)(KT) * (BK * 2) + aoff[U][I], \
                                            smem + (S) * STAGE2 + ((U) * 16 + wave * 2 + (I)) * 1024)
#define FW_PP_W(S, KT, U, I) FW_GLDS16(wbase + (size_t)(KT) * (BK * 2) + woff[U][I], smem + (S) * STAGE2 + TM * BK * 2 + ((U) * 16 + wave * 2 + (I)) * 1024)
#define FW_PP_ISSUE6(S, KT) do { FW_PP_W(S, KT, 0, 0); FW_PP_W(S, KT, 0, 1); FW_PP_W(S, KT, 1, 0); FW_PP_W(S, KT, 1, 1); FW_PP_A(S, KT, 0, 0, tap6); FW_PP_A(S, KT, 0, 1, tap6); \
                                 if (CONV) conv_tap_next(p.cv, tap6); } while (0)
#define FW_PP_ISSUE2(S, KT) do { FW_PP_A(S, KT, 1, 0, tap2); FW_PP_A(S, KT, 1, 1, tap2); if (CONV) conv_tap_next(p.cv, tap2); } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Round 2 (against round 1's gemm_bf16_pp_kernel, same slot table): what changed was read off the ISA:
//   * the loop is peeled by hand (first / steady / second-last / last slab as compile-time flags), so a phase is ONE basic block;
//   * the accumulators are pinned by an empty asm after every burst, so no MFMA drifts across the barrier that follows.
// +2..5 % on the DiT shapes (profiles/r02/gemm_experiments.md).  Tried on this skeleton and measured WITHOUT gain, hence not kept:
// LDS-DMA issued from the LOAD phases instead of from inside the bursts (+-0), the slot barrier signalled one k-step early so its
// release latency runs under the last MFMAs (+-0), a software L2 prefetch 3-4 slabs ahead (-10 %: the extra line requests cost
// more than the misses they hide), a rotated K start per work-group against channel hot-spotting (-6 %: it breaks the lockstep L2
// sharing of A bands / W panels), staggered work-group starts to de-phase the residual epilogues (-3..-20 % before the epilogue rewrite; +-0.3 % after it, with the XCDs offset by 0.5 / 1 / 2 us in the first round: the residual epilogue is bound by one CU's loads in flight, not by the chip-wide burst).  The TIMING build
// (TS, tools/gemm_timeline.py) stamps s_memtime at the phase boundaries.
// ---------------------------------------------------------------------------------------------------------------
__device__ unsigned long long g_gemm_ts[2 * 64];     // TIMING build: [group][slab 0..3][phase 0..3][start | end of work]
// TIMING build, per work-group (tile): s_memrealtime (100 MHz) at entry, after the prologue barrier, at the end of the mainloop, at
// the end of the epilogue (stores issued), after the stores have drained; [5] = HW_ID (which CU ran it).  tools/gemm_timeline.py
// rebuilds the per-CU timeline from it: prologue / mainloop / epilogue per tile and the gap between consecutive tiles on a CU.
constexpr int TILE_TS_MAX = 8192;
__device__ unsigned long long g_gemm_tile_ts[TILE_TS_MAX * 6];

template <int TS, bool CONV>     // TS: 0 = product, 1 = TIMING build (phase + tile stamps), 2 = tile stamps only (does not perturb the loop)
__global__ __launch_bounds__(512, 2) void gemm_bf16_four_slot_kernel(GemmArgs p) {
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
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GM;
        const int gsz = min(p.tiles_m - first_m, GM);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[2][2], woff[2][2];       // [unit][piece]; CONV: aoff holds the packed output pixel of the row instead
    int achunk[2][2];
    ConvTap tap6 = {0, 0, 0, 0}, tap2 = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = u * 128 + (wave * 2 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            if (CONV) { aoff[u][i] = conv_row_pack(p.cv, min(m0 + row, p.M - 1)); achunk[u][i] = chunk * 16; }
            else aoff[u][i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
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

    const bool tile_ts = (TS == 1 || TS == 2) && wave == 0 && lane == 0 && blockIdx.x < TILE_TS_MAX;
    if (tile_ts) {
        g_gemm_tile_ts[blockIdx.x * 6 + 0] = __builtin_amdgcn_s_memrealtime();
        g_gemm_tile_ts[blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    }
    const int nk = p.K / BK;       // >= 4 (launcher)
    FW_PP_ISSUE6(0, 0); FW_PP_ISSUE2(0, 0);
    FW_PP_ISSUE6(1, 1); FW_PP_ISSUE2(1, 1);
    fw_wait_vm<8>();
    FW_BARRIER();
    if (tile_ts) g_gemm_tile_ts[blockIdx.x * 6 + 1] = __builtin_amdgcn_s_memrealtime();
    if (grp == 1) FW_BARRIER();

    int kt = 0;
    unsigned ballast = 0;     // TS == 3 only
    auto slab = [&](auto first_tag, auto has1_tag, auto has2_tag, auto st_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, HAS1 = decltype(has1_tag)::value, HAS2 = decltype(has2_tag)::value;
        // Round 3 experiment, measured and NOT kept: the steady loop unrolled by the two LDS stages (stage = compile-time constant,
        // fragment-read addresses without v_add_u32) ran 9 % SLOWER (1186 vs 1293 TF/s on qkv, profiles/r03/gemm_valu_probe.txt: more
        // address registers live, 164 B of scratch), and the VALU ballast probe (TS == 3, FW_GEMM_VAR bit 8: +16 VALU instructions per
        // slab) costs nothing at all: VALU issue is free in this loop -- unlike in the attention kernels, whose VALU cycles add to
        // their matrix cycles.  st_tag is always "run time" (-1).
        constexpr int STC = decltype(st_tag)::value;
        const int st = STC >= 0 ? STC : (kt & 1);
        const char* base = smem + st * STAGE2;
        // TIMING build: s_memtime at the start (barrier passed) and at the end of the work of every phase, slabs 16..19, work-group 0
        const bool ts_on = TS == 1 && blockIdx.x == 0 && wn == 0 && kt >= 16 && kt < 20;
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
            if (TS == 3 && ks == 1) {      // sensitivity probe: 16 plain VALU instructions of ballast per slab (8 here, 8 in MFMA1)
#pragma unroll
                for (int b8 = 0; b8 < 8; ++b8) asm volatile("v_add_u32 %0, %0, 1" : "+v"(ballast));
            }
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
            if (TS == 3 && ks == 1) {
#pragma unroll
                for (int b8 = 0; b8 < 8; ++b8) asm volatile("v_add_u32 %0, %0, 1" : "+v"(ballast));
            }
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
    using SR = std::integral_constant<int, -1>;
    slab(T{}, T{}, T{}, SR{});
    while (kt < nk - 2) slab(F{}, T{}, T{}, SR{});
    slab(F{}, T{}, F{}, SR{});
    slab(F{}, F{}, F{}, SR{});
    if (grp == 0) FW_BARRIER();
    if (TS == 3) asm volatile("" :: "v"(ballast));
    if (tile_ts) g_gemm_tile_ts[blockIdx.x * 6 + 2] = __builtin_amdgcn_s_memrealtime();
    epilogue_256(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
    if (TS == 1 || TS == 2) {
        if (tile_ts) g_gemm_tile_ts[blockIdx.x * 6 + 3] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tile_ts) g_gemm_tile_ts[blockIdx.x * 6 + 4] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256x64 tile with FOUR waves, one per SIMD with the whole 512-entry register file (wave tile 128 x 128 = 16 accumulators in
// the accumulation half), 128-B LDS rows in two 64 KiB stages -- the shape of the vendor library's kernel for these GEMMs
// (rocprofv3 on torch.matmul: a hand-written MT256x256x64 kernel, 256 threads, 130 KiB LDS, 1.43-1.58 PF on this box).  Against
// gemm_bf16_w4_kernel (half-slab ring, compiler-ordered): full 128-B rows for the LDS-DMA, ONE barrier per 64 MFMAs, and an
// instruction order pinned by hand with sched_barrier: per k-step 16 MFMAs with the 8 fragment reads of the NEXT k-step and the
// LDS-DMA pieces dealt out between them, so no wait ever covers reads that were issued just before it.
//   after the barrier of slab T (slab T+1 landed, every wave done reading slab T):
//     P3: MFMA k-step 3 of slab T   | read frags(T+1, k0) | DMA 8 pieces of slab T+2 -> stage of T
//     P0: MFMA k-step 0 of slab T+1 | read frags(T+1, k1) | DMA the other 8 pieces
//     P1: MFMA k-step 1             | read frags(T+1, k2)
//     P2: MFMA k-step 2             | read frags(T+1, k3) ; lgkmcnt(0), vmcnt(0), s_barrier
// Per wave and slab: 64 MFMAs (2048 cycles), 32 ds_read_b128, 16 DMA pieces; LDS reads per slab and CU 128 KiB (8-wave kernels: 192).
// ---------------------------------------------------------------------------------------------------------------
#define FW_NOMOVE() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256, 1) void gemm_bf16_w4b_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];

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

    // DMA pieces (1 KiB = 8 rows x 128 B): wave w streams A pieces 8w..8w+7 (tile rows 64w .. 64w+63) and the same W pieces.
    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[8], woff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        aoff[i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
        woff[i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
    }
#define FW_4B_A(S, KT, I) FW_GLDS16(abase + (size_t)(KT) * (BK * 2) + aoff[I], smem + (S) * STAGE2 + (wave * 8 + (I)) * 1024)
#define FW_4B_W(S, KT, I) FW_GLDS16(wbase + (size_t)(KT) * (BK * 2) + woff[I], smem + (S) * STAGE2 + TM * BK * 2 + (wave * 8 + (I)) * 1024)

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int a_addr[4], b_addr[4];              // byte offsets of this lane's fragment chunk of k-step ks in stage 0 (+ rb * 4096)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = ((2 * ks + hi) ^ swz) << 4;
        a_addr[ks] = (wm * 128 + fi) * 128 + c;
        b_addr[ks] = TM * BK * 2 + (wn * 128 + fi) * 128 + c;
    }

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t fa[2][4], fb[2][4];           // [set][row / column block]

    const int nk = p.K / BK;               // >= 4 (launcher)
#pragma unroll
    for (int i = 0; i < 8; ++i) { FW_GLDS16(abase + aoff[i], smem + (wave * 8 + i) * 1024); FW_GLDS16(wbase + woff[i], smem + TM * BK * 2 + (wave * 8 + i) * 1024); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { FW_GLDS16(abase + BK * 2 + aoff[i], smem + STAGE2 + (wave * 8 + i) * 1024); FW_GLDS16(wbase + BK * 2 + woff[i], smem + STAGE2 + TM * BK * 2 + (wave * 8 + i) * 1024); }
    fw_wait_vm<16>();
    FW_BARRIER();

    // One k-step: 16 MFMAs on fragment set CUR, with (READ) the 8 fragment reads of k-step `rks` of stage `rst` into the other set
    // after MFMAs 0..7 and (NDMA pieces, first one = piece `p0`) LDS-DMA pieces of slab `dk` into stage `dst` dealt out between
    // the MFMAs; nothing may be reordered.  A global_load_lds_dwordx4 occupies the CU's one texture-address path for 16 cycles
    // (1 KiB at 64 B/clk) and BLOCKS the issuing wave until it is accepted: four waves issuing their pieces together serialise
    // behind each other (~60 cycles each, matrix pipe idle -- measured: +1000 cycles per slab, exactly 64 pieces x 16 cycles).  So
    // the pieces are spread over three k-steps (one per 3 MFMAs) and the waves take DIFFERENT MFMA slots (slot = 3 i + wave % 3)
    // -- which, measured, changes nothing either way (1100 TF/s spread or bunched): the cost follows the BYTES, see docs/kernels.md.
    const int dslot = wave % 3;
    auto kstep = [&](auto cur_tag, auto read_tag, auto dma_tag, int rst, int rks, int dst, int dk) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr bool READ = decltype(read_tag)::value;
        constexpr int GB = decltype(dma_tag)::value;            // -1: no DMA in this k-step; else first of its 16 global MFMA slots (0 / 16 / 32)
        const char* rbase = smem + rst * STAGE2;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[CUR][rb], fb[CUR][nb], acc[rb][nb], 0, 0, 0);
                const int j = rb * 4 + nb;                      // 0..15
                if (READ && j < 8) {                            // a0 b0 a1 b1 .. : the last read is 8 MFMAs old at the next k-step
                    FW_NOMOVE();
                    if (j & 1) fb[CUR ^ 1][j >> 1] = *(const bf16x8_t*)(rbase + b_addr[rks] + (j >> 1) * 4096);
                    else fa[CUR ^ 1][j >> 1] = *(const bf16x8_t*)(rbase + a_addr[rks] + (j >> 1) * 4096);
                    FW_NOMOVE();
                }
                if (GB >= 0) {                                  // the 48 MFMA slots of P3, P0, P1 carry the 16 pieces: piece g / 3 in the
                    const int g = GB + j;                       // wave's own slot (g % 3 == dslot) of every MFMA triple
                    FW_NOMOVE();
                    if (g % 3 == dslot) {
                        const int i = g / 3;                    // 0..15: A pieces 0..7, then W pieces 0..7
                        if (i < 8) { FW_4B_A(dst, dk, i); } else { FW_4B_W(dst, dk, i - 8); }
                    }
                    FW_NOMOVE();
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using DN = std::integral_constant<int, -1>;
    using D0 = std::integral_constant<int, 0>;
    using D16 = std::integral_constant<int, 16>;
    using D32 = std::integral_constant<int, 32>;

    // prologue: slab 0 k-steps 0..2 (no DMA: stage 0 is still being read), then the barrier that opens slab 1
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        fa[0][rb] = *(const bf16x8_t*)(smem + a_addr[0] + rb * 4096);
        fb[0][rb] = *(const bf16x8_t*)(smem + b_addr[0] + rb * 4096);
    }
    __builtin_amdgcn_s_setprio(1);
    kstep(S0{}, T{}, DN{}, 0, 1, 0, 0);
    kstep(S1{}, T{}, DN{}, 0, 2, 0, 0);
    kstep(S0{}, T{}, DN{}, 0, 3, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fw_wait_vm<0>();
    FW_BARRIER();

    // steady state: T = slab whose k-step 3 is pending in set 1; slab T+1 landed in stage (T+1)&1; slab T+2 goes to stage T&1
    int Tk = 0;
    auto body = [&](auto has2_tag) {
        constexpr bool HAS2 = decltype(has2_tag)::value;
        const int sn = (Tk + 1) & 1, so = Tk & 1;
        if (HAS2) {
            kstep(S1{}, T{}, D0{}, sn, 0, so, Tk + 2);          // P3: MFMA slots  0..15
            kstep(S0{}, T{}, D16{}, sn, 1, so, Tk + 2);         // P0: MFMA slots 16..31
            kstep(S1{}, T{}, D32{}, sn, 2, so, Tk + 2);         // P1: MFMA slots 32..47
        } else {
            kstep(S1{}, T{}, DN{}, sn, 0, so, 0);
            kstep(S0{}, T{}, DN{}, sn, 1, so, 0);
            kstep(S1{}, T{}, DN{}, sn, 2, so, 0);
        }
        kstep(S0{}, T{}, DN{}, sn, 3, so, 0);                   // P2: a k-step without DMA lets the last pieces land
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fw_wait_vm<0>();
        FW_BARRIER();
        ++Tk;
    };
    while (Tk < nk - 2) body(T{});
    body(F{});                                                  // Tk = nk-2: slab nk-1 k-steps 0..2, no more DMA
    kstep(S1{}, F{}, DN{}, 0, 0, 0, 0);                         // k-step 3 of the last slab
    __builtin_amdgcn_s_setprio(0);
    FW_BARRIER();
    epilogue_w4(p, smem, acc, wave, wm, wn, fi, hi, lane, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_bf16_w4d_kernel (round 6 experiment, FW_GEMM_KERNEL=8): gemm_bf16_w4b_kernel with its steady-state LDS-DMA pieces issued as
// buffer_load ... lds through an SGPR descriptor instead of global_load_lds with a 64-bit address pair.  The vendor library's kernel
// for these shapes (Custom_Cijk_..._MT256x256x64_MI16x16x1, disassembled from torch's hipblaslt code object: four waves, 128 x
// v_mfma_f32_16x16x32_bf16 + 32 ds_read_b128 + 16 `buffer_load_dwordx4 ... offen lds` + 3 s_barrier per slab, no ds_write) IS a
// four-wave LDS-DMA kernel and reaches 1500 TF/s: what blocked w4b's waves is worth re-measuring on this request form.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4d_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];

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

    // DMA pieces (1 KiB = 8 rows x 128 B): wave w streams A pieces 8w..8w+7 (tile rows 64w .. 64w+63) and the same W pieces.
    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[8], woff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        aoff[i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
        woff[i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
    }
    // the pieces of the steady loop: buffer_load ... lds through an SGPR descriptor (the form the vendor's kernel uses; rows past M / N
    // are out of range of the descriptor), one 32-bit offset register per operand and piece parity, the rest in the scalar offset
    const long long arem = (long long)(p.M - m0) * p.lda * 2, wrem = (long long)(p.N - n0) * p.ldw * 2;
    const auto ars = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0,
                                                       (int)(unsigned)(arem > 0xffffffffLL ? 0xffffffffLL : arem), 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw), 0,
                                                       (int)(unsigned)(wrem > 0xffffffffLL ? 0xffffffffLL : wrem), 0x00020000);
    const int pr = lane >> 3, pc = lane & 7;
    const int va[2] = {pr * (int)p.lda * 2 + ((pc ^ (pr >> 1)) << 4), pr * (int)p.lda * 2 + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int vw[2] = {pr * (int)p.ldw * 2 + ((pc ^ (pr >> 1)) << 4), pr * (int)p.ldw * 2 + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int astep = __builtin_amdgcn_readfirstlane((int)p.lda * 16), wstep = __builtin_amdgcn_readfirstlane((int)p.ldw * 16);
#define FW_4D_A(S, KT, I) __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, FW_LDS_PTR(smem + (S) * STAGE2 + (wave * 8 + (I)) * 1024), 16, va[(I) & 1], (KT) * (BK * 2) + (wave * 8 + (I)) * astep, 0, 0)
#define FW_4D_W(S, KT, I) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, FW_LDS_PTR(smem + (S) * STAGE2 + TM * BK * 2 + (wave * 8 + (I)) * 1024), 16, vw[(I) & 1], (KT) * (BK * 2) + (wave * 8 + (I)) * wstep, 0, 0)

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int a_addr[4], b_addr[4];              // byte offsets of this lane's fragment chunk of k-step ks in stage 0 (+ rb * 4096)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = ((2 * ks + hi) ^ swz) << 4;
        a_addr[ks] = (wm * 128 + fi) * 128 + c;
        b_addr[ks] = TM * BK * 2 + (wn * 128 + fi) * 128 + c;
    }

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t fa[2][4], fb[2][4];           // [set][row / column block]

    const int nk = p.K / BK;               // >= 4 (launcher)
#pragma unroll
    for (int i = 0; i < 8; ++i) { FW_GLDS16(abase + aoff[i], smem + (wave * 8 + i) * 1024); FW_GLDS16(wbase + woff[i], smem + TM * BK * 2 + (wave * 8 + i) * 1024); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { FW_GLDS16(abase + BK * 2 + aoff[i], smem + STAGE2 + (wave * 8 + i) * 1024); FW_GLDS16(wbase + BK * 2 + woff[i], smem + STAGE2 + TM * BK * 2 + (wave * 8 + i) * 1024); }
    fw_wait_vm<16>();
    FW_BARRIER();

    // One k-step: 16 MFMAs on fragment set CUR, with (READ) the 8 fragment reads of k-step `rks` of stage `rst` into the other set
    // after MFMAs 0..7 and (NDMA pieces, first one = piece `p0`) LDS-DMA pieces of slab `dk` into stage `dst` dealt out between
    // the MFMAs; nothing may be reordered.  A global_load_lds_dwordx4 occupies the CU's one texture-address path for 16 cycles
    // (1 KiB at 64 B/clk) and BLOCKS the issuing wave until it is accepted: four waves issuing their pieces together serialise
    // behind each other (~60 cycles each, matrix pipe idle -- measured: +1000 cycles per slab, exactly 64 pieces x 16 cycles).  So
    // the pieces are spread over three k-steps (one per 3 MFMAs) and the waves take DIFFERENT MFMA slots (slot = 3 i + wave % 3)
    // -- which, measured, changes nothing either way (1100 TF/s spread or bunched): the cost follows the BYTES, see docs/kernels.md.
    const int dslot = wave % 3;
    auto kstep = [&](auto cur_tag, auto read_tag, auto dma_tag, int rst, int rks, int dst, int dk) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr bool READ = decltype(read_tag)::value;
        constexpr int GB = decltype(dma_tag)::value;            // -1: no DMA in this k-step; else first of its 16 global MFMA slots (0 / 16 / 32)
        const char* rbase = smem + rst * STAGE2;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[CUR][rb], fb[CUR][nb], acc[rb][nb], 0, 0, 0);
                const int j = rb * 4 + nb;                      // 0..15
                if (READ && j < 8) {                            // a0 b0 a1 b1 .. : the last read is 8 MFMAs old at the next k-step
                    FW_NOMOVE();
                    if (j & 1) fb[CUR ^ 1][j >> 1] = *(const bf16x8_t*)(rbase + b_addr[rks] + (j >> 1) * 4096);
                    else fa[CUR ^ 1][j >> 1] = *(const bf16x8_t*)(rbase + a_addr[rks] + (j >> 1) * 4096);
                    FW_NOMOVE();
                }
                if (GB >= 0) {                                  // the 48 MFMA slots of P3, P0, P1 carry the 16 pieces: piece g / 3 in the
                    const int g = GB + j;                       // wave's own slot (g % 3 == dslot) of every MFMA triple
                    if (g % 3 == 0) {                           // compile-time slot (every third MFMA): no per-slot branch
                        const int i = g / 3;                    // 0..15: A pieces 0..7, then W pieces 0..7
                        FW_NOMOVE();
                        if (i < 8) { FW_4D_A(dst, dk, i); } else { FW_4D_W(dst, dk, i - 8); }
                        FW_NOMOVE();
                    }
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using DN = std::integral_constant<int, -1>;
    using D0 = std::integral_constant<int, 0>;
    using D16 = std::integral_constant<int, 16>;
    using D32 = std::integral_constant<int, 32>;

    // prologue: slab 0 k-steps 0..2 (no DMA: stage 0 is still being read), then the barrier that opens slab 1
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        fa[0][rb] = *(const bf16x8_t*)(smem + a_addr[0] + rb * 4096);
        fb[0][rb] = *(const bf16x8_t*)(smem + b_addr[0] + rb * 4096);
    }
    __builtin_amdgcn_s_setprio(1);
    kstep(S0{}, T{}, DN{}, 0, 1, 0, 0);
    kstep(S1{}, T{}, DN{}, 0, 2, 0, 0);
    kstep(S0{}, T{}, DN{}, 0, 3, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fw_wait_vm<0>();
    FW_BARRIER();

    // steady state: T = slab whose k-step 3 is pending in set 1; slab T+1 landed in stage (T+1)&1; slab T+2 goes to stage T&1
    int Tk = 0;
    auto body = [&](auto has2_tag) {
        constexpr bool HAS2 = decltype(has2_tag)::value;
        const int sn = (Tk + 1) & 1, so = Tk & 1;
        if (HAS2) {
            kstep(S1{}, T{}, D0{}, sn, 0, so, Tk + 2);          // P3: MFMA slots  0..15
            kstep(S0{}, T{}, D16{}, sn, 1, so, Tk + 2);         // P0: MFMA slots 16..31
            kstep(S1{}, T{}, D32{}, sn, 2, so, Tk + 2);         // P1: MFMA slots 32..47
        } else {
            kstep(S1{}, T{}, DN{}, sn, 0, so, 0);
            kstep(S0{}, T{}, DN{}, sn, 1, so, 0);
            kstep(S1{}, T{}, DN{}, sn, 2, so, 0);
        }
        kstep(S0{}, T{}, DN{}, sn, 3, so, 0);                   // P2: a k-step without DMA lets the last pieces land
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fw_wait_vm<0>();
        FW_BARRIER();
        ++Tk;
    };
    while (Tk < nk - 2) body(T{});
    body(F{});                                                  // Tk = nk-2: slab nk-1 k-steps 0..2, no more DMA
    kstep(S1{}, F{}, DN{}, 0, 0, 0, 0);                         // k-step 3 of the last slab
    __builtin_amdgcn_s_setprio(0);
    FW_BARRIER();
    epilogue_w4(p, smem, acc, wave, wm, wn, fi, hi, lane, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_bf16_w4c_kernel (round 6 experiment, FW_GEMM_KERNEL=7): the four-wave kernel above with the global -> LDS path the vendor
// library's kernel for these shapes uses -- buffer_load_dwordx4 into VGPRs, ds_write_b128 into the stage -- instead of LDS-DMA.
// Why: gemm_bf16_w4b_kernel's finding is that a global_load_lds BLOCKS the issuing wave until the texture-address path accepts it
// (+1000 cycles per slab with one wave per SIMD and nobody else to issue MFMAs meanwhile); an ordinary VMEM load returns its issue slot
// at once and the ds_write costs an LDS issue like a fragment read.  One wave per SIMD has the registers for it: 256 accumulators in the
// accumulation half, 64 staging registers (this wave's 16 pieces of one slab), 64 fragment registers.
//   body(T) (after the barrier of slab T: slab T+1 in stage (T+1)&1, everyone done reading stage T&1; stg[] = slab T+2, requested a slab ago):
//     P3: MFMA k-step 3 of slab T   | read frags(T+1, k0) | every 3rd MFMA: ds_write stg[i] -> stage T&1, then stg[i] <- load slab T+3
//     P0: MFMA k-step 0 of slab T+1 | read frags(T+1, k1) |   (16 pieces over the 48 MFMA slots of P3, P0, P1)
//     P1: MFMA k-step 1             | read frags(T+1, k2) |
//     P2: MFMA k-step 2             | read frags(T+1, k3) ; lgkmcnt(0) (own writes done), s_barrier -- NO vmcnt wait: the loads of slab
//                                                           T+3 stay in flight across the barrier and are waited for piece by piece
// Same LDS image, fragment reads, k order and epilogue as w4b: bit-identical results.  Rows past M / N: buffer descriptor range (zeros).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4c_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];

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
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int gid = wg / per_group;
        const int first_m = gid * GM;
        const int gsz = min(p.tiles_m - first_m, GM);
        const int in_g = wg - gid * per_group;
        tm = first_m + in_g % gsz;
        tn = in_g / gsz;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    const long long arem = (long long)(p.M - m0) * p.lda * 2, wrem = (long long)(p.N - n0) * p.ldw * 2;
    const auto ars = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0,
                                                       (int)(unsigned)(arem > 0xffffffffLL ? 0xffffffffLL : arem), 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw), 0,
                                                       (int)(unsigned)(wrem > 0xffffffffLL ? 0xffffffffLL : wrem), 0x00020000);
    // a piece = 8 rows x 128 B: lane -> row pr, chunk pc; the swizzle chunk ^ ((row >> 1) & 7) sits on the SOURCE address (even / odd
    // piece), so the LDS destination is lane-linear like the DMA's: piece base + lane * 16
    const int pr = lane >> 3, pc = lane & 7;
    const int va[2] = {pr * (int)p.lda * 2 + ((pc ^ (pr >> 1)) << 4), pr * (int)p.lda * 2 + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int vw[2] = {pr * (int)p.ldw * 2 + ((pc ^ (pr >> 1)) << 4), pr * (int)p.ldw * 2 + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int astep = __builtin_amdgcn_readfirstlane((int)p.lda * 16), wstep = __builtin_amdgcn_readfirstlane((int)p.ldw * 16);
    const int pq = wave * 8;                   // this wave's pieces 8w .. 8w+7 of both operands (tile rows 64w .. 64w+63)
#define FW_4C_LDA(KT, I) __builtin_amdgcn_raw_buffer_load_b128(ars, va[(I) & 1], (KT) * (BK * 2) + (pq + (I)) * astep, 0)
#define FW_4C_LDW(KT, I) __builtin_amdgcn_raw_buffer_load_b128(wrs, vw[(I) & 1], (KT) * (BK * 2) + (pq + (I)) * wstep, 0)
    char* const wr_base = smem + pq * 1024 + lane * 16;      // + stage * STAGE2 (+ TM * BK * 2 for W) + piece * 1024

    // prologue slabs 0 and 1 by LDS-DMA (as in w4b: nothing to overlap with yet)
    const char* abase = (const char*)(p.A + (int64_t)m0 * p.lda);
    const char* wbase = (const char*)(p.W + (int64_t)n0 * p.ldw);
    unsigned aoff[8], woff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        aoff[i] = (unsigned)(min(row, p.M - 1 - m0) * (int)p.lda + chunk * 8) * 2u;
        woff[i] = (unsigned)(min(row, p.N - 1 - n0) * (int)p.ldw + chunk * 8) * 2u;
    }

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int a_addr[4], b_addr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = ((2 * ks + hi) ^ swz) << 4;
        a_addr[ks] = (wm * 128 + fi) * 128 + c;
        b_addr[ks] = TM * BK * 2 + (wn * 128 + fi) * 128 + c;
    }

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t fa[2][4], fb[2][4];
    u32x4_t stg[16];                           // staging: A pieces 0..7, W pieces 0..7 of ONE slab

    const int nk = p.K / BK;                   // >= 4 (launcher)
#pragma unroll
    for (int i = 0; i < 8; ++i) { FW_GLDS16(abase + aoff[i], smem + (wave * 8 + i) * 1024); FW_GLDS16(wbase + woff[i], smem + TM * BK * 2 + (wave * 8 + i) * 1024); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { FW_GLDS16(abase + BK * 2 + aoff[i], smem + STAGE2 + (wave * 8 + i) * 1024); FW_GLDS16(wbase + BK * 2 + woff[i], smem + STAGE2 + TM * BK * 2 + (wave * 8 + i) * 1024); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { stg[i] = FW_4C_LDA(2, i); stg[8 + i] = FW_4C_LDW(2, i); }      // slab 2 into the staging registers
    // (the BUILTIN, not inline asm: the compiler's wait-count pass reads it, learns that the LDS-DMA of slab 0 is done and keeps exact
    //  per-register counts for the staged loads -- with the asm form it assumed LDS-DMA still pending at the loop head and put a
    //  vmcnt(0) in front of the first fragment read of every slab, which waits for the loads requested one k-step earlier)
    __builtin_amdgcn_s_waitcnt(0x8F70);        // vmcnt(32): slab 0 landed (slab 1's DMA and slab 2's loads stay in flight)
    FW_BARRIER();

    // One k-step: 16 MFMAs on fragment set CUR; after MFMAs 0..7 the 8 fragment reads of k-step `rks` of stage `rst` into the other
    // set; PIECES: at every third MFMA slot g = GB + j one piece i = g / 3 -- ds_write of stg[i] into stage `dst`, then (LOADS) the
    // request of the same piece of slab `dk` into stg[i].  Nothing may be reordered across the fences.
    auto kstep = [&](auto cur_tag, auto read_tag, auto dma_tag, auto loads_tag, int rst, int rks, int dst, int dk) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr bool READ = decltype(read_tag)::value;
        constexpr int GB = decltype(dma_tag)::value;            // -1: no pieces in this k-step; else first of its 16 global MFMA slots (0 / 16 / 32)
        constexpr bool LOADS = decltype(loads_tag)::value;
        const char* rbase = smem + rst * STAGE2;
        char* wbase_l = wr_base + dst * STAGE2;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[CUR][rb], fb[CUR][nb], acc[rb][nb], 0, 0, 0);
                const int j = rb * 4 + nb;                      // 0..15
                if (READ && j < 8) {
                    FW_NOMOVE();
                    if (j & 1) fb[CUR ^ 1][j >> 1] = *(const bf16x8_t*)(rbase + b_addr[rks] + (j >> 1) * 4096);
                    else fa[CUR ^ 1][j >> 1] = *(const bf16x8_t*)(rbase + a_addr[rks] + (j >> 1) * 4096);
                    FW_NOMOVE();
                }
                if (GB >= 0 && (GB + j) % 3 == 0) {
                    const int i = (GB + j) / 3;                 // 0..15: A pieces 0..7, then W pieces 0..7
                    FW_NOMOVE();
                    *(u32x4_t*)(wbase_l + (i < 8 ? i * 1024 : TM * BK * 2 + (i - 8) * 1024)) = stg[i];
                    if (LOADS) stg[i] = i < 8 ? FW_4C_LDA(dk, i) : FW_4C_LDW(dk, i - 8);
                    FW_NOMOVE();
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using DN = std::integral_constant<int, -1>;
    using D0 = std::integral_constant<int, 0>;
    using D16 = std::integral_constant<int, 16>;
    using D32 = std::integral_constant<int, 32>;

    // prologue: slab 0 k-steps 0..2, then the barrier that opens slab 1
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        fa[0][rb] = *(const bf16x8_t*)(smem + a_addr[0] + rb * 4096);
        fb[0][rb] = *(const bf16x8_t*)(smem + b_addr[0] + rb * 4096);
    }
    __builtin_amdgcn_s_setprio(1);
    kstep(S0{}, T{}, DN{}, F{}, 0, 1, 0, 0);
    kstep(S1{}, T{}, DN{}, F{}, 0, 2, 0, 0);
    kstep(S0{}, T{}, DN{}, F{}, 0, 3, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x4F70);        // vmcnt(16): slab 1 landed (the 16 loads of slab 2 stay in flight)
    FW_BARRIER();

    int Tk = 0;
    auto body = [&](auto has2_tag, auto has3_tag) __attribute__((always_inline)) {
        constexpr bool HAS2 = decltype(has2_tag)::value, HAS3 = decltype(has3_tag)::value;
        const int sn = (Tk + 1) & 1, so = Tk & 1;
        if (HAS2) {
            kstep(S1{}, T{}, D0{}, has3_tag, sn, 0, so, Tk + 3);        // P3: MFMA slots  0..15
            kstep(S0{}, T{}, D16{}, has3_tag, sn, 1, so, Tk + 3);       // P0: MFMA slots 16..31
            kstep(S1{}, T{}, D32{}, has3_tag, sn, 2, so, Tk + 3);       // P1: MFMA slots 32..47
        } else {
            kstep(S1{}, T{}, DN{}, F{}, sn, 0, so, 0);
            kstep(S0{}, T{}, DN{}, F{}, sn, 1, so, 0);
            kstep(S1{}, T{}, DN{}, F{}, sn, 2, so, 0);
        }
        kstep(S0{}, T{}, DN{}, F{}, sn, 3, so, 0);                      // P2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // own ds_writes (slab T+2) and fragment reads done; loads stay in flight
        FW_BARRIER();
        ++Tk;
    };
    while (Tk < nk - 3) body(T{}, T{});
    body(T{}, F{});                                                     // Tk = nk-3: writes slab nk-1, nothing left to request
    body(F{}, F{});                                                     // Tk = nk-2: slab nk-1 k-steps 0..2
    kstep(S1{}, F{}, DN{}, F{}, 0, 0, 0, 0);                            // k-step 3 of the last slab
    __builtin_amdgcn_s_setprio(0);
    FW_BARRIER();
    epilogue_w4(p, smem, acc, wave, wm, wn, fi, hi, lane, m0, n0);
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
    // M-tiles per group of the ping-pong kernel's tile order (FW_GEMM_VAR >> 4 overrides, A/B).  Interleaved A/B on the forward's
    // shapes (tools/gemm_ab.py, profiles/r02/gemm_experiments.md): 4 against 8 is +1.6 % on the N = 5120 gate + residual GEMMs,
    // +6.8 % on the bicross output projection, +-0 on qkv / ffn0, and -2..3 % on the narrow VGGT outputs (N = 3072 / 4096); 2, 3, 5, 6
    // lose to 4 and 16 / 32 lose 10-15 %.
    p.group_m = (fw_get_option(FW_OPT_GEMM_VAR) >> 4) ? (fw_get_option(FW_OPT_GEMM_VAR) >> 4) : ((N + TN - 1) / TN >= 20 ? 4 : GROUP_M);
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
    if (big && K >= 4 * BK) {
        // FW_GEMM_KERNEL: 4 (default) = ping-pong kernel; 5 = four-wave kernel (independent implementation, agreement tests)
        const int kern = fw_get_option(FW_OPT_GEMM_KERNEL);
        // Short M tail (VGGT: 32865 rows = 128 full row bands + 97 rows): a 129th band of 256-row tiles costs a whole extra
        // round of the grid (516 tiles on 256 CUs = 3 rounds for 2.02 rounds of work).  Peel it: the full bands go to the 256x256
        // kernel (512 tiles = 2 rounds), the <= 128 leftover rows to the 128x128 kernel in a second, tiny launch.  Same
        // k-order per output element in both kernels, disjoint output rows.
        const int tail = M % TM;
        if (tail > 0 && tail <= BM && M >= 2 * TM) {
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
            hipLaunchKernelGGL((gemm_bf16_kernel<false, 4>), dim3((unsigned)t.tiles_n), dim3(256), 0, (hipStream_t)stream, t);      // one row of tiles: deep ring
            return (int)hipGetLastError();
        }
        p.tiles_m = (M + TM - 1) / TM; p.tiles_n = (N + TN - 1) / TN;
        const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
        if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_bf16: grid too large"); return FW_E_BADARG; }
        hipStream_t st = (hipStream_t)stream;
        // (the four-wave kernel used to serve K <= 1280: it measured +8..12 % there -- because its epilogue did not carry the per-element
        // activation switch the ping-pong kernel's did.  With that gone the ping-pong kernel is 8-15 % faster at K = 1024 / 1152 as
        // well (tools/gemm_ab.py --short-k), so the four-wave kernel is now only the independent implementation behind FW_GEMM_KERNEL=5.)
        // 6 ... = the two-slot ping-pong kernels of round 4 (gemm_pp.hip)
        if (kern == 5) {
            hipLaunchKernelGGL(gemm_bf16_w4b_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
        } else if (kern == 8 && p.lda < (1 << 21) && p.ldw < (1 << 21)) {      // round-6 experiment: four waves, descriptor LDS-DMA
            hipLaunchKernelGGL(gemm_bf16_w4d_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
        } else if (kern == 7 && p.lda < (1 << 21) && p.ldw < (1 << 21)) {      // round-6 experiment: four waves, VGPR-staged loads
            hipLaunchKernelGGL(gemm_bf16_w4c_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
        } else if (kern >= 6 && fw_launch_gemm_pp(p, kern, fw_get_option(FW_OPT_GEMM_VAR), st)) {
        } else if (fw_get_option(FW_OPT_GEMM_VAR) & 2) {
            hipLaunchKernelGGL((gemm_bf16_four_slot_kernel<1, false>), dim3((unsigned)nwg), dim3(512), 0, st, p);      // TIMING build
        } else if (fw_get_option(FW_OPT_GEMM_VAR) & 8) {
            hipLaunchKernelGGL((gemm_bf16_four_slot_kernel<3, false>), dim3((unsigned)nwg), dim3(512), 0, st, p);      // run-time-stage loop + 16 VALU of ballast per slab
        } else if (fw_get_option(FW_OPT_GEMM_VAR) & 4) {
            hipLaunchKernelGGL((gemm_bf16_four_slot_kernel<2, false>), dim3((unsigned)nwg), dim3(512), 0, st, p);      // tile stamps only (run-time-stage loop)
        } else {
            hipLaunchKernelGGL((gemm_bf16_four_slot_kernel<0, false>), dim3((unsigned)nwg), dim3(512), 0, st, p);
        }
        return (int)hipGetLastError();
    }
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_bf16: grid too large"); return FW_E_BADARG; }
    if (nwg <= 256) hipLaunchKernelGGL((gemm_bf16_kernel<false, 4>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, 2>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int fw_conv_gemm_bf16(const uint16_t* x, int64_t ldx, int C, int T, int H, int W, int kt, int kh, int kw,
                                 int sh, int sw, int ph, int pw, int up, int t0, int nt,
                                 const uint16_t* Wt, int64_t ldw, void* Cout, int64_t ldc, int out_dtype, int N,
                                 const float* bias, int act, const float* g1, const float* g0,
                                 const void* res, int64_t ldr, int res_dtype, void* stream) {
    if (N <= 0 || nt <= 0) return 0;
    if (C <= 0 || (C % BK) || (ldx % 8) || (ldw % 8) || (((uintptr_t)x) & 15) || (((uintptr_t)Wt) & 15)) {
        fw_set_error("fw_conv_gemm_bf16: C must be a positive multiple of 64, ldx / ldw % 8 == 0, x / W 16-byte aligned"); return FW_E_BADARG; }
    if (T < 1 || H < 1 || W < 1 || kt < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || (up != 1 && up != 2) ||
        t0 < 0 || t0 + nt > T) { fw_set_error("fw_conv_gemm_bf16: bad geometry"); return FW_E_BADARG; }
    if (out_dtype != FW_DT_BF16 && out_dtype != FW_DT_F32) { fw_set_error("fw_conv_gemm_bf16: bad out_dtype"); return FW_E_BADARG; }
    if (res_dtype != FW_DT_NONE && res == nullptr) { fw_set_error("fw_conv_gemm_bf16: res_dtype set but res NULL"); return FW_E_BADARG; }
    const int Hu = H * up, Wu = W * up;
    const int Ho = (Hu + 2 * ph - kh) / sh + 1, Wo = (Wu + 2 * pw - kw) / sw + 1;
    if (Ho < 1 || Wo < 1) { fw_set_error("fw_conv_gemm_bf16: empty output map"); return FW_E_BADARG; }
    // the kernels keep (frame | yo*sh | xo*sw) of a row in 8 + 12 + 12 bits
    if (T > 255 || (Ho - 1) * sh > 4095 || (Wo - 1) * sw > 4095) {
        fw_set_error("fw_conv_gemm_bf16: T <= 255 and (Ho-1)*sh, (Wo-1)*sw <= 4095 required (chunk the volume)"); return FW_E_BADARG; }
    const int64_t M64 = (int64_t)nt * Ho * Wo;
    const int64_t K64 = (int64_t)kt * kh * kw * C;
    if (M64 > 0x7fffffff || K64 > 0x7fffffff || ldw < K64) { fw_set_error("fw_conv_gemm_bf16: M or K too large, or ldw < taps * C"); return FW_E_BADARG; }
    GemmArgs p;
    p.A = x; p.lda = ldx; p.W = Wt; p.ldw = ldw; p.C = Cout; p.ldc = ldc; p.out_dtype = out_dtype;
    p.M = (int)M64; p.N = N; p.K = (int)K64; p.bias = bias; p.act = act; p.g1 = g1; p.g0 = g0;
    p.res = res; p.ldr = ldr; p.res_dtype = res ? res_dtype : FW_DT_NONE;
    p.group_m = GROUP_M;
    p.cv.T = T; p.cv.H = H; p.cv.W = W; p.cv.Ho = Ho; p.cv.Wo = Wo; p.cv.kt = kt; p.cv.kh = kh; p.cv.kw = kw;
    p.cv.sh = sh; p.cv.sw = sw; p.cv.ph = ph; p.cv.pw = pw; p.cv.ups = up == 2 ? 1 : 0; p.cv.t0 = t0; p.cv.cpt = C / BK;
    const int M = p.M, K = p.K;
    // same tile rule as fw_gemm_bf16 (no tail peel: a ragged last band of a feature map with 10^5 .. 10^7 rows costs nothing)
    bool big = M >= 2048 && (N >= 1024 || (N >= 256 && (int64_t)(M / TM) * ((N + TN - 1) / TN) >= 512));
    const int forced = fw_get_option(FW_OPT_GEMM_TILE);
    if (forced == 128) big = false;
    if (forced == 256) big = true;
    const uintptr_t cmask = (out_dtype == FW_DT_F32) ? 15 : 7;
    const uintptr_t rmask = (res_dtype == FW_DT_F32) ? 15 : 7;
    if ((N % 4) || (ldc % 4) || (((uintptr_t)Cout) & cmask) || (res && ((ldr % 4) || (((uintptr_t)res) & rmask)))) big = false;
    if ((((uintptr_t)bias) | ((uintptr_t)g1) | ((uintptr_t)g0)) & 15) big = false;
    hipStream_t st = (hipStream_t)stream;
    if (big && K >= 4 * BK) {
        p.tiles_m = (M + TM - 1) / TM; p.tiles_n = (N + TN - 1) / TN;
        hipLaunchKernelGGL((gemm_bf16_four_slot_kernel<0, true>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), 0, st, p);
        return (int)hipGetLastError();
    }
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    if (nwg > 0x7fffffff) { fw_set_error("fw_conv_gemm_bf16: grid too large"); return FW_E_BADARG; }
    hipLaunchKernelGGL((gemm_bf16_kernel<true, 2>), dim3((unsigned)nwg), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

// Measurement hook (tools/gemm_timeline.py): the phase timestamps written by the TIMING build of the ping-pong kernel.
extern "C" int fw_debug_gemm_timestamps(unsigned long long* host_out, int n) {
    // n <= 128: the phase stamps; n = 128 + 6 * tiles: followed by the per-tile stamps of the first `tiles` work-groups
    if (n <= 0 || n > 128 + 6 * TILE_TS_MAX) { fw_set_error("fw_debug_gemm_timestamps: n out of range"); return FW_E_BADARG; }
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_ts), sizeof(unsigned long long) * (n < 128 ? n : 128));
    if (rc || n <= 128) return rc;
    return (int)hipMemcpyFromSymbol(host_out + 128, HIP_SYMBOL(g_gemm_tile_ts), sizeof(unsigned long long) * (n - 128));
}

extern "C" int fw_gemv_f32(const float* x, const float* W, int64_t ldw, const float* bias, float* out,
                           int N, int K, int act_in_silu, int act_out, void* stream) {
    if (N <= 0 || K <= 0) { fw_set_error("fw_gemv_f32: bad shape"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, W, ldw, bias, out, N, K, act_in_silu, act_out);
    return (int)hipGetLastError();
}

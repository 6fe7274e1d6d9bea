// Two-slot ping-pong bf16 GEMM kernels for gfx950 (round 4).  Same argument block, tile order, LDS image and fused epilogue as the
// four-slot kernel in gemm.hip; see the block comment of each kernel for what differs.
#include "gemm_common.h"

using namespace fwgemm;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the ping-pong kernel with TWO slots per k-slab instead of four (gemm_bf16_two_slot_kernel below is its final form).
//
// What the phase timeline of gemm_bf16_four_slot_kernel (gemm.hip) says (profiles/r02/gemm_timeline_call5.txt): a slab costs ~3100 cycles
// for 2048 cycles of MFMA; its four barrier-separated slots last 780 / 704 / 672 / 988 cycles against 512 of matrix work each: every
// slot pays the barrier turn-around (the matrix pipe of a SIMD is idle from the last MFMA of one group until the first of the other)
// and the LDS-DMA instructions issued from inside the bursts stall their wave with nobody else on the SIMD issuing MFMAs.  Here
//   * a wave does all 32 MFMAs of a slab in ONE burst (1024 cycles): the A fragments of wave-tile rows 64..127 are read during
//     the first 16 MFMAs into the registers whose rows 0..63 fragments have just been consumed (no extra registers, no barrier:
//     the data landed a slab ago and nobody overwrites it before the next barrier) -- 2 barriers per slab, not 4;
//   * the LDS-DMA pieces go through buffer_load ... lds with an SGPR descriptor: one 32-bit offset register per operand and piece
//     parity instead of a 64-bit address pair per piece, the k advance and the piece's row offset live in the scalar offset, rows
//     past M / N are out of range of the descriptor (zeros, no clamps).
//     slot:        2t          2t+1         2t+2         2t+3
//     group A:   LOAD(t)     MFMA(t)      LOAD(t+1)    MFMA(t+1)          LOAD(t): B(t) fragments + A rows 0..63 of the wave tile;
//     group B:   MFMA(t-1)   LOAD(t)      MFMA(t)      LOAD(t+1)          MFMA(t) re-reads A rows 64..127 of the same stage
// First form (gemm_bf16_pp3_kernel, two 64 KiB stages, measured and removed; profiles/r04/gemm_ab_call3_two_slot_kernel.txt): B(t+2)
// and A0(t+2) can only be requested in slot 2t+2, so part of the pieces stays inside group B's bursts; with 0 / 2 / 4 of a wave's 4
// A0 pieces moved to group A's LOAD phase it ran 1262 / 1317 / 1337 TF/s on the qkv shape (four-slot kernel: 1279) -- the more
// pieces leave the bursts, the faster.  Same k order per output element as the other kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------------------------
#define FW_BLDS16(rs, voff, soff, ldsptr) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, FW_LDS_PTR(ldsptr), 16, voff, soff, 0, 0)

// ---------------------------------------------------------------------------------------------------------------
// gemm_bf16_two_slot_kernel: the two-slot kernel with EVERY LDS-DMA piece issued from a LOAD phase and a third W stage.
//
// gemm_bf16_pp3_kernel's measurements (profiles/r04/gemm_ab_call3_two_slot_kernel.txt): the more pieces leave the MFMA bursts the
// faster it runs (A0A = 0 / 2 / 4: 1262 / 1317 / 1337 TF/s on the qkv shape).  What keeps the rest inside the bursts there is the
// two-stage LDS image: B(t+2) and A0(t+2) may only be requested in slot 2t+2, when group A is loading and group B computing.  The
// 160 KiB of LDS hold one more 32 KiB W stage (A: 2 x 32 KiB, W: 3 x 32 KiB): B(t) lives in stage t % 3, so B(t+3) may be requested
// as soon as B(t) has been read (slot 2t+1) -- three to four slots (1.5-2 slab times) ahead -- and the requests balance over the
// two groups' LOAD phases, 8 pieces per wave and slab each:
//     group A, LOAD(t) (slot 2t):     A0(t+1) -> A stage (t+1) & 1,   W rows   0..127 of slab t+2 -> W stage (t+2) % 3
//     group B, LOAD(t) (slot 2t+1):   A1(t+1) -> A stage (t+1) & 1,   W rows 128..255 of slab t+2 -> W stage (t+2) % 3
//   (B(t-1), the previous tenant of W stage (t+2) % 3, was last read in slot 2t-1; A0(t-1) in slot 2t-1, A1(t-1) in slot 2t.)
// The bursts are 32 MFMAs + the 8 fragment re-reads, nothing else.  Waits (slab t, each ahead of the barrier before the first read):
//     group A, end of MFMA(t):  vmcnt(4)   own A0(t+1), W-lo(t+1) landed; W-lo(t+2) stays in flight
//     group B, end of LOAD(t):  vmcnt(8)   own W-hi(t+1) landed (group A reads it in slot 2t+2); A1(t+1), W-hi(t+2) stay in flight
//     group B, end of MFMA(t):  vmcnt(4)   own A1(t+1) landed; W-hi(t+2) stays in flight
// TS = 1: s_memtime stamps at the phase boundaries of slabs 16..19 of work-group 0 (fw_debug_gemm_pp_timestamps, tools/gemm_timeline.py).
// ---------------------------------------------------------------------------------------------------------------
constexpr int P4_A_STAGE = TM * BK * 2;                 // 32 KiB
constexpr int P4_W_BASE = 2 * P4_A_STAGE;               // 64 KiB
constexpr int P4_W_STAGE = TN * BK * 2;                 // 32 KiB
constexpr int P4_LDS = P4_W_BASE + 3 * P4_W_STAGE;      // 160 KiB
__device__ unsigned long long g_pp_ts[2 * 4 * 8];      // TIMING build: [group][slab 16..19][stamp 0..5]

// Measured and NOT kept (profiles/r04/gemm_ab_call5_early_barrier_is_slower.txt): executing the barrier that ends a burst 4 / 8 / 12 MFMAs
// before the burst's end, so that the other group starts while this one's last MFMAs drain (the phase timeline shows 120-180 cycles per
// hand-over): -10 % in all three forms (1247-1252 against 1387 TF/s) -- two waves of a SIMD issuing MFMAs at once is worse than the gap.
// (Also measured and not kept: the accumulators pinned in the accumulation half of the register file through asm MFMAs with "+a"
//  operands -- bit-identical, -0.7 %, profiles/r04/gemm_ab_call10_agpr_accumulators_no_gain.txt.)
template <int TS>
__global__ __launch_bounds__(512, 2) void gemm_bf16_two_slot_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[P4_LDS];

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

    const long long arem = (long long)(p.M - m0) * p.lda * 2, wrem = (long long)(p.N - n0) * p.ldw * 2;
    const auto ars = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0,
                                                       (int)(unsigned)(arem > 0xffffffffLL ? 0xffffffffLL : arem), 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw), 0,
                                                       (int)(unsigned)(wrem > 0xffffffffLL ? 0xffffffffLL : wrem), 0x00020000);
    const int pr = lane >> 3, pc = lane & 7;
    const int va[2] = {pr * (int)p.lda * 2 + ((pc ^ (pr >> 1)) << 4), pr * (int)p.lda * 2 + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int vw[2] = {pr * (int)p.ldw * 2 + ((pc ^ (pr >> 1)) << 4), pr * (int)p.ldw * 2 + ((pc ^ ((pr >> 1) | 4)) << 4)};
    const int astep = __builtin_amdgcn_readfirstlane((int)p.lda * 16), wstep = __builtin_amdgcn_readfirstlane((int)p.ldw * 16);
    // this wave's 4 A pieces and 4 W pieces of every slab: tile rows (grp * 128 + 32 wn) .. + 31 of both operands
    const int q0 = grp * 16 + 4 * wn;
#define FW_P4_A(AS, KT, J) FW_BLDS16(ars, va[(J) & 1], (KT) * (BK * 2) + (q0 + (J)) * astep, smem + (AS) * P4_A_STAGE + (q0 + (J)) * 1024)
#define FW_P4_W(WS, KT, J) FW_BLDS16(wrs, vw[(J) & 1], (KT) * (BK * 2) + (q0 + (J)) * wstep, smem + P4_W_BASE + (WS) * P4_W_STAGE + (q0 + (J)) * 1024)
    auto issue_a = [&](int as, int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) FW_P4_A(as, kt, j);
    };
    auto issue_w = [&](int ws, int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) FW_P4_W(ws, kt, j);
    };

    const int fi = lane & 31, hi = lane >> 5;
    const int swz = (fi >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;
    const int a_row_off = (grp * 128 + fi) * 128;
    const int b_row_off = P4_W_BASE + (wn * 64 + fi) * 128;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t afr[2][4], bfr[2][4];

    const int nk = p.K / BK;       // >= 4 (launcher)

    // ---- prologue: A(0), B(0), A(1), B(1), B(2): every wave its own pieces, in the order the steady-state waits assume ----
    issue_a(0, 0); issue_w(0, 0);
    issue_a(1, 1); issue_w(1, 1);
    issue_w(2, 2);
    fw_wait_vm<12>();
    FW_BARRIER();
    if (grp == 1) FW_BARRIER();

    int kt = 0;
    int ws = 0;                    // W stage of slab kt = kt % 3
    auto slab = [&](auto g_tag, auto first_tag, auto n1_tag, auto n2_tag) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, N1 = decltype(n1_tag)::value, N2 = decltype(n2_tag)::value;   // kt == 0; kt + 1 < nk; kt + 2 < nk
        const int as = kt & 1;
        const char* abase = smem + as * P4_A_STAGE;
        const char* bbase = smem + ws * P4_W_STAGE;
        const bool ts_on = TS == 1 && blockIdx.x == 0 && wn == 0 && kt >= 16 && kt < 20;
        auto stamp = [&](int which) __attribute__((always_inline)) {
            if (ts_on) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) g_pp_ts[(G * 4 + (kt - 16)) * 8 + which] = t;
            }
        };
        stamp(0);
        // ---------------- LOAD(kt): B fragments + A rows 0..63 of the wave tile, then this wave's 8 pieces
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bfr[0][ks] = *(const bf16x8_t*)(bbase + b_row_off + coff[ks]);
            bfr[1][ks] = *(const bf16x8_t*)(bbase + b_row_off + 32 * 128 + coff[ks]);
            afr[0][ks] = *(const bf16x8_t*)(abase + a_row_off + coff[ks]);
            afr[1][ks] = *(const bf16x8_t*)(abase + a_row_off + 32 * 128 + coff[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!FIRST && N1) {
            issue_a(as ^ 1, kt + 1);                          // A0 / A1 of slab kt+1 -> the A stage slab kt-1 has left
            if (N2) issue_w(ws == 0 ? 2 : ws - 1, kt + 2);    // W half of slab kt+2 -> W stage (kt+2) % 3 = (kt-1) % 3
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (G == 1 && N1) {                                   // own W-hi(kt+1) landed (group A reads it in the next slot)
            if (FIRST) fw_wait_vm<4>(); else if (N2) fw_wait_vm<8>(); else fw_wait_vm<4>();
        }
        stamp(2);
        FW_BARRIER();
        stamp(3);
        // ---------------- MFMA(kt)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            afr[0][ks] = *(const bf16x8_t*)(abase + a_row_off + 64 * 128 + coff[ks]);
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            afr[1][ks] = *(const bf16x8_t*)(abase + a_row_off + 96 * 128 + coff[ks]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[0][ks], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0][ks], bfr[1][ks], acc[2][1], 0, 0, 0);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[0][ks], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[1][ks], bfr[1][ks], acc[3][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        asm volatile("" : "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        stamp(4);
        if (N1) {                                             // own A pieces (and group A: W-lo) of slab kt+1 landed; W(kt+2) stays in flight
            if (N2) fw_wait_vm<4>(); else fw_wait_vm<0>();
        }
        stamp(5);
        FW_BARRIER();
        ++kt;
        ws = ws == 2 ? 0 : ws + 1;
    };
    using T = std::true_type;
    using F = std::false_type;
    auto mainloop = [&](auto g_tag) __attribute__((always_inline)) {
        // (distinct opaque markers at both ends of an arm: without them the compiler hoists the common first LOAD above the group
        //  branch and merges the arms' tails, and then carries 16 fragment registers through scratch around one arm's loop)
        asm volatile("; two-slot mainloop, group %0: begin" ::"n"(decltype(g_tag)::value) : "memory");
        slab(g_tag, T{}, T{}, T{});
        while (kt < nk - 2) slab(g_tag, F{}, T{}, T{});
        slab(g_tag, F{}, T{}, F{});                           // kt = nk - 2
        slab(g_tag, F{}, F{}, F{});                           // kt = nk - 1
        asm volatile("; two-slot mainloop, group %0: end" ::"n"(decltype(g_tag)::value) : "memory");
    };
    if (grp == 0) { mainloop(std::integral_constant<int, 0>{}); FW_BARRIER(); }
    else mainloop(std::integral_constant<int, 1>{});
    epilogue_256(p, smem, acc, wave, grp, wn, fi, hi, lane, m0, n0);
}

}  // namespace

bool fw_launch_gemm_pp(const GemmArgs& p, int kern, int var, hipStream_t st) {
    if (kern != 9 || !(p.lda < (1 << 21) && p.ldw < (1 << 21))) return false;           // 32-bit byte offsets inside a tile
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(512);
    if (var & 2) hipLaunchKernelGGL((gemm_bf16_two_slot_kernel<1>), grid, block, 0, st, p);      // TIMING build
    else hipLaunchKernelGGL((gemm_bf16_two_slot_kernel<0>), grid, block, 0, st, p);
    return true;
}

// Measurement hook (tools/gemm_pp_timeline.py): the phase stamps written by the TIMING build of gemm_bf16_two_slot_kernel.
extern "C" int fw_debug_gemm_pp_timestamps(unsigned long long* host_out, int n) {
    if (n <= 0 || n > 64) { fw_set_error("fw_debug_gemm_pp_timestamps: n out of range"); return FW_E_BADARG; }
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_ts), sizeof(unsigned long long) * n);
}

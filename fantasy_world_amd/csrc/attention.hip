// Fused non-causal attention forward (flash style, fp32 online softmax) for gfx950, head_dim 64/96/128.
//
// Work-group = 8 waves (512 threads) = 256 query rows of one (batch, head); each wave owns 32 query rows.
// Per 64-key tile:
//   S^T[key][q]  = K_tile * Q^T      "swapped" QK^T: v_mfma_f32_32x32x16_bf16 with A = K rows, B = Q rows, so the
//                                    accumulator column (lane&31) is the QUERY: a lane owns one query's scores and
//                                    the running max / sum are lane-local (one cross-half exchange per tile).
//   O^T[d][q]   += Vt_tile * P^T     A = Vt rows (d), B = P^T taken straight from the score registers.
// The B-operand k-slot (lane>>5, j) of the PV MFMA receives the score register 8*s+j, whose key index inside a
// 32-key block is swap_bits_2_3(16*s + 8*(lane>>5) + j) (32x32 C/D layout: row = (r&3)+8*(r>>2)+4*(lane>>5)).
// fw_v_transpose bakes exactly that permutation into Vt's key axis, so no cross-lane shuffle is needed between
// the two MFMAs and every LDS fragment read is a conflict-free ds_read_b128:
//   K tile  : 64 rows x 256 B (row stride fixed at 256 B, hd/8 valid 16-B chunks), chunk ^= row&15
//   Vt tile : hd rows x 128 B, chunk ^= (row>>1)&7
// Both tiles are filled with global_load_lds_dwordx4 (swizzle applied on the source address) into 2-slot rings, one
// barrier per tile.  The loop is software pipelined: K runs one tile ahead of V, and the QK^T MFMAs of tile t+1 are
// issued between the row-max and the exponentials of tile t, so matrix and vector work of one wave overlap.  Work-groups are ordered so that one XCD works on one head at a time (K/V stay in its L2).
#include "fw_common.h"
#include <type_traits>

namespace {

constexpr int QB = 256;      // query rows per work-group
constexpr int KVB = 64;      // keys per tile
constexpr int K_TILE_BYTES = KVB * 256;

struct AttnArgs {
    const uint16_t* Q; int64_t ldq, bsq;
    const uint16_t* K; int64_t ldk, bsk;
    const uint16_t* Vt; int64_t lkp;
    uint16_t* O; int64_t ldo, bso;
    int batch, heads, Lq, Lk;
    float scale_log2;   // softmax scale * log2(e)
    int accumulate;
    int nqb;            // q-blocks per (batch, head)
    // split-KV (attention_sp_kernel only; part == nullptr: off): the key range is cut into `nsplit` runs of `tiles_per_split`
    // 64-key tiles, one work-group per (batch, head, q-block, run); a work-group writes its UN-normalised output, its row sums
    // and its softmax shifts (fp32, [bh][qb][run][256 rows][HD + 2]) and attention_combine_kernel merges the runs.
    float* part;
    int nsplit, tiles_per_split;
    int nofast;         // A/B (FW_ATTN_VAR bit 10): 1 = tile requests in pointer form (64-bit multiply-add + ragged-row fix-up), not by descriptor
    int prio;           // experiment (FW_ATTN_VAR bits 8-9): 1 = s_setprio 1 for waves 4..7, 2 = for waves 0..3, before the tile loop;
                        // measured +-0 on every shape (profiles/r03/microbench_attention_static_priority.txt), default 0
};

template <int HD>
__global__ __launch_bounds__(512, 2) void attention_kernel(AttnArgs p) {
    constexpr int KS = HD / 16;          // k-steps of the QK^T contraction
    constexpr int DB = HD / 32;          // 32-row blocks of O^T
    constexpr int NCH = HD / 8;          // valid 16-B chunks per K row
    constexpr int VT_TILE_BYTES = HD * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * VT_TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 31, hi = lane >> 5;

    // ---- work-group -> (batch*head, q-block): XCD x owns a contiguous range of (bh, qb) items ------------
    int item;
    {
        const int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bh = item / p.nqb;
    const int qb = item - bh * p.nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;

    const uint16_t* Qp = p.Q + (int64_t)b * p.bsq + (int64_t)h * HD;
    const uint16_t* Kp = p.K + (int64_t)b * p.bsk + (int64_t)h * HD;
    const uint16_t* Vp = p.Vt + ((int64_t)b * p.heads + h) * HD * p.lkp;
    uint16_t* Op = p.O + (int64_t)b * p.bso + (int64_t)h * HD;

    // ---- Q fragments (B operand): lane (fi,hi) holds Q[q0+fi][16*ks + 8*hi .. +7] -------------------------
    const int q_row = qb * QB + wave * 32 + fi;
    bf16x8_t qf[KS];
    {
        const int qr = min(q_row, p.Lq - 1);
        const uint16_t* src = Qp + (int64_t)qr * p.ldq + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(src + ks * 16);
    }

    // ---- staging addresses --------------------------------------------------------------------------------
    // K tile: 16 pieces of 1 KiB (4 rows x 256 B); wave w issues pieces w and w+8.
    //   lane -> row = 4*pc + lane/16, physical chunk = lane%16, logical chunk = phys ^ (row&15) (skip if >= NCH)
    const uint16_t* kg[2];
    bool kvalid[2];
    int krow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pc = wave + 8 * i;
        const int row = pc * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ (row & 15);
        krow[i] = row;
        kvalid[i] = chunk < NCH;
        kg[i] = Kp + chunk * 8;
    }
    // Vt tile: HD/8 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w, w+8 (if < HD/8).
    //   lane -> row d = 8*pc + lane/8, physical chunk = lane%8, logical chunk = phys ^ ((d>>1)&7)
    const uint16_t* vg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pc = wave + 8 * i;
        const int d = pc * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((d >> 1) & 7);
        vg[i] = Vp + (int64_t)min(d, HD - 1) * p.lkp + chunk * 8;
    }

    // K ring: 2 slots of K_TILE_BYTES; Vt ring: 2 slots of VT_TILE_BYTES
    auto stage_k = [&](int slot, int t) {
        char* k_lds = smem + slot * K_TILE_BYTES;
        const int k0 = t * KVB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (kvalid[i]) {
                const int kr = min(k0 + krow[i], p.Lk - 1);
                FW_GLDS16(kg[i] + (int64_t)kr * p.ldk, k_lds + (wave + 8 * i) * 1024);
            }
        }
    };
    auto stage_v = [&](int slot, int t) {
        char* v_lds = smem + 2 * K_TILE_BYTES + slot * VT_TILE_BYTES;
        const int k0 = t * KVB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + 8 * i;
            if (pc < HD / 8) FW_GLDS16(vg[i] + k0, v_lds + pc * 1024);
        }
    };

    // ---- fragment read offsets ------------------------------------------------------------------------------
    int kcoff[KS];                    // K: chunk (2ks+hi) ^ (fi&15)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kcoff[ks] = fi * 256 + (((2 * ks + hi) ^ (fi & 15)) << 4);
    int vcoff[4];                     // Vt: chunk (2s+hi) ^ ((fi>>1)&7)
#pragma unroll
    for (int s = 0; s < 4; ++s) vcoff[s] = 2 * K_TILE_BYTES + fi * 128 + (((2 * s + hi) ^ ((fi >> 1) & 7)) << 4);

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1.0e30f;    // running max of raw scores (finite sentinel: no inf-inf)
    float l_run = 0.f;         // this half-wave's partial row sum
    const float c = p.scale_log2;
    const int nt = (p.Lk + KVB - 1) / KVB;
    const bool ragged = (p.Lk & (KVB - 1)) != 0;

    // S^T = K Q^T for the K tile in ring slot `slot` (two 32-key blocks), masked if it is the ragged last tile
    auto qk = [&](f32x16_t& s0, f32x16_t& s1, int slot, int t) {
        const char* base = smem + slot * K_TILE_BYTES;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t k0f = *(const bf16x8_t*)(base + kcoff[ks]);
            bf16x8_t k1f = *(const bf16x8_t*)(base + 32 * 256 + kcoff[ks]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0f, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1f, qf[ks], s1, 0, 0, 0);
        }
        if (ragged && t == nt - 1) {
            const int kbase = t * KVB + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kbase + (r & 3) + 8 * (r >> 2);
                if (kk >= p.Lk) s0[r] = -1.0e30f;
                if (kk + 32 >= p.Lk) s1[r] = -1.0e30f;
            }
        }
    };

    // One KV tile: online softmax of the scores already in (c0,c1), QK^T of the NEXT tile into (n0,n1) issued
    // between the max and the exponentials (independent MFMA work that overlaps the softmax VALU), then P V.
    auto tile = [&](f32x16_t& c0, f32x16_t& c1, f32x16_t& n0, f32x16_t& n1, int t) {
        if (t + 2 < nt) stage_k(t & 1, t + 2);
        if (t + 1 < nt) stage_v((t + 1) & 1, t + 1);
        float mx = fmaxf(c0[0], c1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(c0[r], c1[r]));
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            m_run = m_new;
        }
        if (t + 1 < nt) qk(n0, n1, (t + 1) & 1, t + 1);
        const float mc = m_run * c;
        uint32_t pw[16];   // P^T packed to bf16: words 0..7 from key block 0, 8..15 from key block 1
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float a0 = __builtin_amdgcn_exp2f(fmaf(c0[r], c, -mc));
            const float a1 = __builtin_amdgcn_exp2f(fmaf(c0[r + 1], c, -mc));
            const float b0 = __builtin_amdgcn_exp2f(fmaf(c1[r], c, -mc));
            const float b1 = __builtin_amdgcn_exp2f(fmaf(c1[r + 1], c, -mc));
            ls += (a0 + a1) + (b0 + b1);
            pw[r >> 1] = pack_bf16x2(a0, a1);
            pw[8 + (r >> 1)] = pack_bf16x2(b0, b1);
        }
        l_run += ls;
        const char* vbase = smem + (t & 1) * VT_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4_t pv4 = {pw[4 * s], pw[4 * s + 1], pw[4 * s + 2], pw[4 * s + 3]};
            bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                bf16x8_t vf = *(const bf16x8_t*)(vbase + d * 32 * 128 + vcoff[s]);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
            }
        }
        __syncthreads();   // K(t+2), V(t+1) landed (vmcnt(0)); every wave is done with K(t+1)'s slot reads and V(t)
    };

    f32x16_t sa0, sa1, sb0, sb1;
    stage_k(0, 0);
    stage_v(0, 0);
    if (nt > 1) stage_k(1, 1);
    __syncthreads();
    qk(sa0, sa1, 0, 0);
    __syncthreads();           // K slot 0 may now be refilled
    for (int t = 0; t < nt; t += 2) {
        tile(sa0, sa1, sb0, sb1, t);
        if (t + 1 < nt) tile(sb0, sb1, sa0, sa1, t + 1);
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l --------------------------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < p.Lq) {
        uint16_t* dst = Op + (int64_t)q_row * p.ldo;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = d * 32 + 8 * g + 4 * hi;
                float v0 = o[d][4 * g + 0] * inv, v1 = o[d][4 * g + 1] * inv;
                float v2 = o[d][4 * g + 2] * inv, v3 = o[d][4 * g + 3] * inv;
                u32x2_t* ptr = (u32x2_t*)(dst + col);
                if (p.accumulate) {
                    const u32x2_t old = *ptr;
                    v0 += __uint_as_float(old[0] << 16);
                    v1 += __uint_as_float(old[0] & 0xffff0000u);
                    v2 += __uint_as_float(old[1] << 16);
                    v3 += __uint_as_float(old[1] & 0xffff0000u);
                }
                u32x2_t w = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                *ptr = w;
            }
        }
    }
}


#define FW_ABARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N> __device__ __forceinline__ void fw_await_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------------------------
// Two-segment ping-pong schedule (attention_pp3_kernel below; round 1's four-segment and first two-segment kernels were
// removed in round 2 -- their measurements are kept here because they are why the schedule looks the way it does).
//
// Measured on the four-segment kernel (rocprofv3 PMC, profiles/r01): ~226 VALU instructions and ~1400 active cycles per
// wave-tile against 1024 cycles of MFMA; with four segments per tile the softmax segment (~1400 cycles) was paired with a
// 512-cycle matrix segment twice per tile, i.e. tile time = 1024 + 2 x 1400: VALU and MFMA time ADD instead of overlapping
// (53 % matrix-pipe utilisation at the real 1.9 GHz clock).  Here a wave alternates only two segments,
//     V(t)   read the Vt(t) tile into the fragment registers, online softmax of tile t            (LDS + VALU)
//     MM(t)  PV(t): HD/8 MFMAs, each followed by the ds_read of one K(t+1) fragment into the register the MFMA just
//            released; then QK^T(t+1): HD/8 MFMAs; DMA of K(t+4) and V(t+3) issued from inside   (matrix pipe)
// and the two 4-wave groups run one segment apart, so a SIMD always has one wave in V and one in MM:
//     slot:       2t      2t+1     2t+2
//     group A:   V(t)    MM(t)    V(t+1)          tile time = 2 x max(V, MM) instead of 2 x V + MM, two barriers per tile
//     group B:   MM(t-1) V(t)     MM(t)
// K and Vt tiles live in 4-deep LDS rings (the kernel is register-bound to one work-group per CU, so all 128 KiB are free):
// K(t) is last read in slot 2t, V(t) in slot 2t+1; K(t+4) / V(t+3) are requested in MM(t) (slot >= 2t+1) and first read in
// slots 2t+7 / 2t+6, i.e. ~3 tile times ahead.  Per-wave DMA order is K(0) | K(1) V(0) | K(2) V(1) | ..., 2 instructions
// each, and the waits are counted:  group A: end of V(t): vmcnt(10) -> own K(t+1); end of MM(t): vmcnt(8) -> own V(t+1)
//                                   group B: end of V(t): vmcnt(4)  -> own V(t+1); end of MM(t): vmcnt(10) -> own K(t+2)
// (vmcnt(0) on the last four tiles).  Rescale of O is deferred until the running max grows by more than 2^8 (exact in
// floating point: P <= 256).  VAR bit 0: s_setprio(1) around MM.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fw_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max of the 16 values of an accumulator: 7 v_max3 + 1 v_max in ONE asm statement (no per-op canonicalisation, one boundary pad)
__device__ __forceinline__ float fw_max16(const f32x16_t& v) {
    float r, t;
    asm("v_max3_f32 %0, %2, %3, %4\n\t"
        "v_max3_f32 %1, %5, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %1, %1, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %1, %1, %14, %15\n\t"
        "v_max3_f32 %0, %0, %16, %17\n\t"
        "v_max_f32 %0, %0, %1"
        : "=&v"(r), "=&v"(t)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
          "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    return r;
}

constexpr int ARING = 4;

// segment timestamps of work-group 0 at tile 100 (attention_pp3_kernel<.., VAR & 2>): [wave][8] shader-clock ticks
__device__ unsigned long long g_attn_ts[8 * 8];
#define FW_TS(K) do { if (TIMING && blockIdx.x == 0 && t == 100 && lane == 0) g_attn_ts[wave * 8 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)

// ---------------------------------------------------------------------------------------------------------------
// attention_pp3_kernel: the two-segment ping-pong schedule above with the softmax cut down to what the
// issue slots allow.  Measured (round-1 two-segment variants, hd 128 vs 64): slot time ~ V-segment + 0.8 x MM-segment, i.e. VALU work of
// one wave hardly overlaps the MFMAs of its SIMD partner -- every instruction of either wave costs an issue slot of the
// shared SIMD, so the lever is the instruction COUNT per tile.  Requires FW_ATTN_Q_PRESCALED (scores arrive in the log2
// domain).  Per 64-key tile and lane:
//   * the running max is folded into QK^T: the first MFMA of a score block takes C = splat(-m_run) (a 16-register block that
//     changes only on a rescale) instead of 0, so the accumulator already holds s - m_run: no multiply-add per score;
//   * no per-tile max: P = exp2(s - m_run) is computed directly, and only if a half-row sum exceeds 2^14 (some P >= 2^8; also
//     catches inf/NaN) -- or on the very first tile -- does the wave take the slow path: exact max, m_run += max, rescale O and
//     l, shift the scores, recompute P.  P stays <= 2^14: exact in fp32/bf16 floating point, no accuracy cost.
//   fast path: 32 v_exp + 31 v_add + 16 v_cvt_pk + 1 compare  (vs ~160 VALU before).
// VAR bit 0: DMA issued between the QK^T MFMAs (else in front of the PV MFMAs).
// ---------------------------------------------------------------------------------------------------------------
template <int HD, int VAR>
__global__ __launch_bounds__(512, 2) void attention_pp3_kernel(AttnArgs p) {
    constexpr bool DMA_QK = (VAR & 1) != 0;
    constexpr bool TIMING = (VAR & 2) != 0;
    constexpr bool BUF = (VAR & 8) == 0;       // tile requests by SGPR descriptor (below); bit 3: the round-3 pointer form
    constexpr bool NOPIN = (VAR & 4) != 0;     // experiment: let LLVM sink the softmax next to the PV MFMAs (intra-wave interleave)    // measurement build: s_memtime at the segment boundaries of one tile
    constexpr int KS = HD / 16, DB = HD / 32, NCH = HD / 8, NFR = HD / 8;
    constexpr int VT_TILE_BYTES = HD * 128;
    constexpr int VROWS = HD / 16, VLANES = VROWS * 8;
    constexpr int V_BASE = ARING * K_TILE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[ARING * K_TILE_BYTES + ARING * VT_TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int fi = lane & 31, hi = lane >> 5;

    int item;
    {
        const int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bh = item / p.nqb;
    const int qb = item - bh * p.nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;

    const uint16_t* Qp = p.Q + (int64_t)b * p.bsq + (int64_t)h * HD;
    const char* Kp = (const char*)(p.K + (int64_t)b * p.bsk + (int64_t)h * HD);
    const char* Vp = (const char*)(p.Vt + ((int64_t)b * p.heads + h) * HD * p.lkp);
    uint16_t* Op = p.O + (int64_t)b * p.bso + (int64_t)h * HD;

    const int q_row = qb * QB + wave * 32 + fi;
    bf16x8_t qf[KS];
    {
        const int qr = min(q_row, p.Lq - 1);
        const uint16_t* src = Qp + (int64_t)qr * p.ldq + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(src + ks * 16);
    }

    const int nt = (p.Lk + KVB - 1) / KVB;
    const bool ragged = (p.Lk & (KVB - 1)) != 0;

    unsigned koff[2];
    bool kvalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 8 * i) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ (row & 15);
        kvalid[i] = chunk < NCH;
        koff[i] = (unsigned)(row * (int)p.ldk + chunk * 8) * 2u;
    }
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = min((wave * 2 + i) * VROWS + (lane >> 3), HD - 1);
        const int chunk = (lane & 7) ^ ((d >> 1) & 7);
        voff[i] = (unsigned)(d * (int)p.lkp + chunk * 8) * 2u;
    }
    const size_t k_tile_stride = (size_t)KVB * (size_t)p.ldk * 2;
    // (round 4: requests by SGPR descriptor + scalar tile offset, as in attention_sp_kernel -- see the comment there.  A tile past the
    //  last one is then a legal request -- rows out of the descriptor's range return zeros, into a ring slot nobody reads any more --
    //  so EVERY tile issues its 4 requests and the counted waits are the same constants from the first tile to the last: the three-way
    //  "steady / draining" choice in front of both barriers of a tile and the t + 4 < nt tests, ~35 scalar instructions and ~10
    //  branches of the ~100 per tile in the round-3 ISA, are gone.  Views of 4 GiB or more: the pointer-form instantiation, VAR bit 3.)
    const size_t kbytes = (size_t)p.Lk * (size_t)p.ldk * 2, vbytes = (size_t)HD * (size_t)p.lkp * 2;
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(unsigned)(kbytes > 0xffffffffull ? 0xffffffffull : kbytes), 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(unsigned)(vbytes > 0xffffffffull ? 0xffffffffull : vbytes), 0x00020000);
    auto issue_k = [&](int t) {
        char* k_lds = smem + (t & (ARING - 1)) * K_TILE_BYTES;
        if constexpr (BUF) {
            const unsigned so = (unsigned)t * (unsigned)k_tile_stride;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (kvalid[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, FW_LDS_PTR(k_lds + (wave + 8 * i) * 1024), 16, (int)koff[i], (int)so, 0, 0);
            return;
        }
        const char* kt = Kp + (size_t)t * k_tile_stride;
        const bool last = ragged && t == nt - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned off = koff[i];
            if (last) {     // ragged last tile: rows beyond the last key re-read the last key (their scores are masked)
                const int row = (wave + 8 * i) * 4 + (lane >> 4);
                const int over = row - (p.Lk - 1 - (nt - 1) * KVB);
                if (over > 0) off -= (unsigned)(over * (int)p.ldk) * 2u;
            }
            if (kvalid[i]) FW_GLDS16(kt + off, k_lds + (wave + 8 * i) * 1024);
        }
    };
    auto issue_v = [&](int t) {
        char* v_lds = smem + V_BASE + (t & (ARING - 1)) * VT_TILE_BYTES;
        if constexpr (BUF) {
            const unsigned so = (unsigned)t * (KVB * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (lane < VLANES) __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, FW_LDS_PTR(v_lds + (wave * 2 + i) * (VROWS * 128)), 16, (int)voff[i], (int)so, 0, 0);
            return;
        }
        const char* vt = Vp + (size_t)t * (KVB * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane < VLANES) FW_GLDS16(vt + voff[i], v_lds + (wave * 2 + i) * (VROWS * 128));
    };

    int kcoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kcoff[ks] = fi * 256 + (((2 * ks + hi) ^ (fi & 15)) << 4);
    int vcoff[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) vcoff[s2] = V_BASE + fi * 128 + (((2 * s2 + hi) ^ ((fi >> 1) & 7)) << 4);

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;       // m_run: shift already folded into the scores (log2 domain)
    f32x16_t negm;                        // splat(-m_run): accumulator input of the first QK^T MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    bf16x8_t fr[NFR];
    f32x16_t s0, s1;
    uint32_t pw[16];

    issue_k(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (BUF || j + 1 < nt) issue_k(j + 1);
        if (BUF || j < nt) issue_v(j);
    }
    fw_await_vm<0>();
    FW_ABARRIER();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        fr[2 * ks] = *(const bf16x8_t*)(smem + kcoff[ks]);
        fr[2 * ks + 1] = *(const bf16x8_t*)(smem + 32 * 256 + kcoff[ks]);
    }
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], qf[0], negm, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], qf[0], negm, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * ks], qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * ks + 1], qf[ks], s1, 0, 0, 0);
    }
    if ((p.prio == 1 && grp == 1) || (p.prio == 2 && grp == 0)) __builtin_amdgcn_s_setprio(1);
    if (grp == 1) FW_ABARRIER();

    int t = 0;
    // slot_tag: ring slot of tile t when it is known at compile time (the steady loop is unrolled by the ring depth, round 3: the
    // fragment-read addresses then fold into ds_read immediates), -1 = t & (ARING - 1)
    auto tile = [&](auto slot_tag, auto has1_tag) {
        constexpr int SL = decltype(slot_tag)::value;
        constexpr bool has1 = decltype(has1_tag)::value;
        const int s_cur = SL >= 0 ? SL : (t & (ARING - 1));
        const int s_nxt = SL >= 0 ? ((SL + 1) & (ARING - 1)) : ((t + 1) & (ARING - 1));
        const bool steady = BUF || t + 4 < nt;
        // ------------------------------------------------------------ V(t): Vt fragments -> registers, softmax
        FW_TS(0);
        {
            const char* vb = smem + s_cur * VT_TILE_BYTES;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)       // first half of the Vt fragments; the rest is fetched behind the first PV MFMAs
#pragma unroll
                for (int d = 0; d < DB; ++d) fr[s2 * DB + d] = *(const bf16x8_t*)(vb + d * 32 * 128 + vcoff[s2]);
        }
        if (ragged && t == nt - 1) {
            const int kbase = t * KVB + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kbase + (r & 3) + 8 * (r >> 2);
                if (kk >= p.Lk) s0[r] = -1.0e30f;
                if (kk + 32 >= p.Lk) s1[r] = -1.0e30f;
            }
        }
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float a0 = __builtin_amdgcn_exp2f(s0[r]), a1 = __builtin_amdgcn_exp2f(s0[r + 1]);
            const float b0 = __builtin_amdgcn_exp2f(s1[r]), b1 = __builtin_amdgcn_exp2f(s1[r + 1]);
            ls += (a0 + a1) + (b0 + b1);
            pw[r >> 1] = pack_bf16x2(a0, a1);
            pw[8 + (r >> 1)] = pack_bf16x2(b0, b1);
        }
        if (__any(t == 0 || !(ls <= 16384.0f))) {
            // slow path (first tile, or some probability of this tile left the safe range): exact tile max, move the running max
            float mx = fmaxf(fw_max16(s0), fw_max16(s1));
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float delta = (t == 0) ? mx : fmaxf(mx, 0.f);      // later tiles only ever raise the max
            const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // nothing accumulated yet on tile 0
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m_run;
            ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float a0 = __builtin_amdgcn_exp2f(s0[r] - delta), a1 = __builtin_amdgcn_exp2f(s0[r + 1] - delta);
                const float b0 = __builtin_amdgcn_exp2f(s1[r] - delta), b1 = __builtin_amdgcn_exp2f(s1[r + 1] - delta);
                ls += (a0 + a1) + (b0 + b1);
                pw[r >> 1] = pack_bf16x2(a0, a1);
                pw[8 + (r >> 1)] = pack_bf16x2(b0, b1);
            }
        }
        l_run += ls;
        if (!NOPIN) {   // keep the softmax in THIS segment (LLVM otherwise sinks it behind the barrier, next to the PV MFMAs)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(pw[i]));
            asm volatile("" : "+v"(l_run));
        }
        FW_TS(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!steady) fw_await_vm<0>();
        else if (grp == 0) fw_await_vm<10>();
        else fw_await_vm<4>();
        FW_TS(2);
        FW_ABARRIER();
        FW_TS(3);
        // ------------------------------------------------------------ MM(t): PV(t), K(t+1) fragments, QK^T(t+1)
        if (!DMA_QK) {
            if (BUF || t + 4 < nt) issue_k(t + 4);
            if (BUF || t + 3 < nt) issue_v(t + 3);
        }
        const char* kb = smem + s_nxt * K_TILE_BYTES;
        const char* vb2 = smem + s_cur * VT_TILE_BYTES;
        constexpr int HF = NFR / 2;
        // fragment register block fr[0..NFR): every MFMA is followed by the ds_read that refills a register it (or an earlier
        // MFMA) released, >= HF MFMAs ahead of its consumer:
        //   PV  s2 = 0,1  (fr[0..HF))   + reads Vt fragments HF..NFR   -> fr[HF..NFR)
        //   PV  s2 = 2,3  (fr[HF..NFR)) + reads K(t+1) fragments 0..HF  -> fr[0..HF)
        //   QK  ks < KS/2 (fr[0..HF))   + reads K(t+1) fragments HF..NFR -> fr[HF..NFR)
        //   QK  ks >= KS/2 (fr[HF..NFR))
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            u32x4_t pv4 = {pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
            bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int i = s2 * DB + d;
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], pf, o[d], 0, 0, 0);
                if (i < HF) {
                    const int j = HF + i;              // Vt fragment j = (s2' = j / DB, d' = j % DB)
                    fr[j] = *(const bf16x8_t*)(vb2 + (j % DB) * 32 * 128 + vcoff[j / DB]);
                } else if (has1) {
                    const int j = i - HF;              // K(t+1) fragment j = (key block j & 1, k-step j >> 1)
                    fr[j] = *(const bf16x8_t*)(kb + (j & 1) * 32 * 256 + kcoff[j >> 1]);
                }
            }
        }
        if (has1) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks == 0) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], qf[0], negm, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], qf[0], negm, 0, 0, 0);
                } else {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * ks], qf[ks], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2 * ks + 1], qf[ks], s1, 0, 0, 0);
                }
                if (ks < KS / 2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = HF + 2 * ks + e;
                        fr[j] = *(const bf16x8_t*)(kb + (j & 1) * 32 * 256 + kcoff[j >> 1]);
                    }
                }
                if (DMA_QK) {
                    if (ks == KS / 2 && (BUF || t + 4 < nt)) issue_k(t + 4);
                    if (ks == KS / 2 + 1 && (BUF || t + 3 < nt)) issue_v(t + 3);
                }
            }
            // pin the issue order: MFMA, ds_read, MFMA, ds_read, ...
            if (!NOPIN)
#pragma unroll
            for (int i = 0; i < NFR + HF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (!DMA_QK && !NOPIN) __builtin_amdgcn_sched_group_barrier(0x008, HF, 0);
        }
        FW_TS(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!steady) fw_await_vm<0>();
        else if (grp == 0) fw_await_vm<8>();
        else fw_await_vm<10>();
        FW_TS(5);
        FW_ABARRIER();
        FW_TS(6);
    };
    // (round 3: unrolling THIS loop by the ring depth -- slot_tag 0..3, as in attention_sp_kernel -- measured 5 % SLOWER at hd 96
    //  (4.54-4.60 ms against 4.34-4.37 ms on one box): the two wave groups already run different phases of the loop, and the four
    //  copies cost more than the 7 address instructions per tile they save.  slot_tag stays run time.)
    {
        using SRT = std::integral_constant<int, -1>;
        // (round 4: one straight-line copy of the loop per wave group, the group's counted waits compile-time constants -- 4 branches
        //  and 8 scalar instructions per tile fewer, 32 VGPRs more: 4.026 vs 4.002 ms at hd 96, no gain; one copy stays.)
        for (; t < nt - 1; ++t) tile(SRT{}, std::true_type{});
        tile(SRT{}, std::false_type{});
    }
    if (BUF) fw_await_vm<0>();      // the requests past the last tile still write this work-group's LDS: drain them before it is released
    if (grp == 0) FW_ABARRIER();

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < p.Lq) {
        uint16_t* dst = Op + (int64_t)q_row * p.ldo;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = d * 32 + 8 * g + 4 * hi;
                float v0 = o[d][4 * g + 0] * inv, v1 = o[d][4 * g + 1] * inv;
                float v2 = o[d][4 * g + 2] * inv, v3 = o[d][4 * g + 3] * inv;
                u32x2_t* ptr = (u32x2_t*)(dst + col);
                if (p.accumulate) {
                    const u32x2_t old = *ptr;
                    v0 += __uint_as_float(old[0] << 16);
                    v1 += __uint_as_float(old[0] & 0xffff0000u);
                    v2 += __uint_as_float(old[1] << 16);
                    v3 += __uint_as_float(old[1] & 0xffff0000u);
                }
                u32x2_t w = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                *ptr = w;
            }
        }
    }
}



// MFMA with explicit register classes for the one-wave-per-SIMD variant, which uses the whole 512-entry register file.
// hipcc gives every MFMA of such a kernel its C/D operands in the accumulation half, so the scores would need one
// v_accvgpr_read per exp.  Scores (and the shift they start from) belong in the architectural half, where VALU reads them;
// output accumulators and Q fragments are MFMA-only and live in the accumulation half.  The hazard recogniser does not look
// inside inline asm: callers keep >= 8 MFMAs between these and any VALU read of their results, or insert fw_mfma_drain().
static __device__ __forceinline__ void fw_mfma_s_first(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
static __device__ __forceinline__ void fw_mfma_s_acc(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
static __device__ __forceinline__ void fw_mfma_o_acc(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
static __device__ __forceinline__ void fw_mfma_drain() {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// attention_sp_kernel: single-stream software pipeline on 32-key half tiles.
//
// Why (tools/probes/valu_probe.hip): inside ONE wave's instruction stream, exp + add + cvt + add ride in the shadow of a
// 32-cycle v_mfma_f32_32x32x16_bf16 almost for free (38 cycles per MFMA), whereas VALU work issued by the SIMD's other wave
// stretches the partner's MFMAs to ~50 cycles (segment timeline of the ping-pong kernels).  So every wave interleaves, in
// program order, the softmax of half tile u with the QK^T MFMAs of half tile u+1, and the fragment reads of the next MFMAs
// with the MFMAs that release their registers:
//     stage A(u):  S(u+1) = K_half(u+1) Q^T   (HD/16 MFMAs)  ||  exp / sum / pack of S(u),  Vt fragments of half u
//     stage B(u):  O^T += Vt_half(u) P(u)^T    (HD/16 MFMAs)  ||  K fragments of half u+2,  tile DMA
// A 32-key half tile keeps the live state small (two 16-register score blocks, one 8-fragment register block shared by K and
// Vt), so 8 waves x 32 query rows still fit two waves per SIMD (<= 256 registers); the two waves of a SIMD are NOT staggered:
// each stream alone keeps the matrix pipe ~85 % busy and the pair interleaves at instruction granularity.  One barrier per
// 64-key tile (4-deep K / Vt LDS rings, K(t+3) and Vt(t+2) requested in iteration t, vmcnt(4) before the barrier).
// Log2-domain scores (FW_ATTN_Q_PRESCALED), running max folded into the accumulator input, overflow check per half tile.
// ---------------------------------------------------------------------------------------------------------------
#define NS_OF(V) (((V) & 2) ? 2 : 1)
template <int HD, int VAR>
__global__ __launch_bounds__((VAR & 2) ? 256 : 512, (VAR & 2) ? 1 : 2) void attention_sp_kernel(AttnArgs p) {
    constexpr bool PINNED = (VAR & 1) != 0;   // sched_group_barrier pins on the stage bodies
    // timing ablations (results are wrong by construction; tools/microbench.py only, FW_ATTN_VAR = 256 + bits, hd 128):
    // 4 = no exp, 8 = no fragment reads in the loop, 16 = no tile DMA and no barrier, 32 = no PV MFMAs
    constexpr bool AB_NOEXP = (VAR & 4) != 0, AB_NODS = (VAR & 8) != 0, AB_NODMA = (VAR & 16) != 0;
    constexpr bool AB_NOPV = (VAR & 32) != 0, AB_NOQK = false;
    constexpr bool PAIR = (VAR & 128) != 0;   // one barrier per TWO 64-key tiles (FW_ATTN_VAR 160 + bits)
    // Round 3: the tile loop unrolled by the ring depth, so the ring slot of every K / Vt fragment read is a compile-time constant
    // and folds into the ds_read's immediate offset.  With run-time slots the loop spent 27 of its 108 VALU instructions per 64-key
    // tile (hd 128) on LDS addresses (16 v_add_u32, 9 v_or_b32, 2 v_lshl_add) -- and VALU cycles ADD to the matrix cycles on this
    // SIMD (docs/kernels.md, "attention: the cap").
    constexpr bool UNR = (VAR & 64) != 0;
    // Round 6 (VAR bit 9): the row sums on the MATRIX pipe.  The 32 v_add_f32 per tile and wave were a third of the loop's vector
    // instructions, and on this SIMD vector issue ADDS to matrix time (docs/kernels.md); a v_mfma_f32_16x16x32_bf16 (4 passes) whose A
    // operand is a per-lane pattern of ones sums one 4-register chunk of P^T -- the operand the PV MFMAs read anyway -- into a 4-register
    // accumulator that lives through the loop: 4 short MFMAs (64 matrix cycles) per tile instead of 32 vector adds.  The sums then are
    // those of the bf16-ROUNDED probabilities, i.e. of exactly what the numerator multiplies (the fp32 sums of the unrounded p they
    // replace differ from them by the rounding noise of P, which now cancels between numerator and denominator).  The A pattern: in
    // the 16x16x32 shape lane l supplies B column l & 15, k-block l >> 4, so the P of query fi = l & 31 (lanes fi and fi + 32) sits in
    // column fi & 15 on the k-blocks of parity fi >> 4, next to query fi ^ 16 on the others; output rows 4 (l >> 4) .. + 3 are the ones
    // lane l reads, so row r must sum the k-blocks of the parity of r >> 2 (tools/probes/mfma16_layout_probe.hip pins the lane -> row /
    // column / k-block map of the 16x16 shapes).  The "sum beyond 2^96" test moves behind the loop (one total bounds every half sum).
    constexpr bool MSUM = (VAR & 512) != 0 && NS_OF(VAR) == 1;
    constexpr bool BUF = UNR && (VAR & 256) == 0;   // tile requests by SGPR descriptor + scalar tile offset (below); bit 8: pointer form
    constexpr int NS = (VAR & 2) ? 2 : 1;     // 32-row sub-blocks per wave: 2 = one wave per SIMD owning 64 query rows
    constexpr int NW = 8 / NS;                // waves per work-group (256 query rows either way)
    constexpr int NP = 16 / NW;               // 1 KiB K pieces / Vt pieces each wave requests per tile
    constexpr int KS = HD / 16, DB = HD / 32, NCH = HD / 8;
    constexpr int HF = HD / 16;               // MFMAs per sub-block (and fragments) per stage: KS for QK^T, 2*DB for PV
    constexpr int PPJ = 8 / HF, PREM = 8 - PPJ * HF;   // score pairs pinned behind each QK^T MFMA, and the remainder
    constexpr int VT_TILE_BYTES = HD * 128;
    constexpr int VROWS = HD / 16, VLANES = VROWS * 8;
    constexpr int V_BASE = ARING * K_TILE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[ARING * K_TILE_BYTES + ARING * VT_TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 31, hi = lane >> 5;

    int item;
    {
        const int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int run = 0;
    if (p.part) { run = item % p.nsplit; item /= p.nsplit; }
    const int bh = item / p.nqb;
    const int qb = item - bh * p.nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int key0 = run * p.tiles_per_split * KVB;                                  // first key of this run (0 without split-KV)
    const int Lk = p.part ? min(p.Lk - key0, p.tiles_per_split * KVB) : p.Lk;       // keys this work-group attends to

    const uint16_t* Qp = p.Q + (int64_t)b * p.bsq + (int64_t)h * HD;
    const char* Kp = (const char*)(p.K + (int64_t)b * p.bsk + (int64_t)h * HD + (int64_t)key0 * p.ldk);
    const char* Vp = (const char*)(p.Vt + ((int64_t)b * p.heads + h) * HD * p.lkp + key0);
    uint16_t* Op = p.O + (int64_t)b * p.bso + (int64_t)h * HD;

    bf16x8_t qf[NS][KS];
#pragma unroll
    for (int sb = 0; sb < NS; ++sb) {
        const int q_row = qb * QB + (wave * NS + sb) * 32 + fi;
        const uint16_t* src = Qp + (int64_t)min(q_row, p.Lq - 1) * p.ldq + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[sb][ks] = *(const bf16x8_t*)(src + ks * 16);
    }
    if (NS == 2) {
        // move the Q fragments to the accumulation half here, once: otherwise hipcc keeps them in VGPRs through the prologue
        // and writes each one to a scratch AGPR right in front of the asm MFMA that reads it (a hazard it cannot see)
#pragma unroll
        for (int sb = 0; sb < NS; ++sb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(qf[sb][ks]));
    }
    const int nt = (Lk + KVB - 1) / KVB;
    const bool ragged = (Lk & (KVB - 1)) != 0;

    // tile DMA: a K tile is 16 pieces of 4 key rows, a Vt tile 16 pieces of HD/16 feature rows; wave w requests pieces w + NW*i
    unsigned koff[NP];
    bool kvalid[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = (wave + NW * i) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ (row & 15);
        kvalid[i] = chunk < NCH;
        koff[i] = (unsigned)(row * (int)p.ldk + chunk * 8) * 2u;
    }
    unsigned voff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int d = min((wave + NW * i) * VROWS + (lane >> 3), HD - 1);
        const int chunk = (lane & 7) ^ ((d >> 1) & 7);
        voff[i] = (unsigned)(d * (int)p.lkp + chunk * 8) * 2u;
    }
    const size_t k_tile_stride = (size_t)KVB * (size_t)p.ldk * 2;
    // Round 4: the requests go through buffer_load ... lds with an SGPR descriptor over this (batch, head, run)'s K rows / Vt rows: the
    // per-lane part of the address is the 32-bit offset register, the tile's byte offset is the SCALAR offset (one s_mul_i32), and the
    // rows of a ragged last tile past the last key are out of the descriptor's range (zeros; their scores are masked) -- instead of a
    // 64-bit multiply-add per request and the ragged-row selects (25 SALU + 6 VALU per tile and wave in the round-3 ISA, on a SIMD
    // whose issue port the ~155 non-MFMA instructions per 32 MFMAs already fill).  The pointer form stays, as its own instantiation, for
    // views of 4 GiB or more (and as the A/B arm, FW_ATTN_VAR bit 10).
    const size_t kbytes = (size_t)(p.Lk - key0) * (size_t)p.ldk * 2, vbytes = ((size_t)HD * (size_t)p.lkp - (size_t)key0) * 2;
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)(unsigned)(kbytes > 0xffffffffull ? 0xffffffffull : kbytes), 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)(unsigned)(vbytes > 0xffffffffull ? 0xffffffffull : vbytes), 0x00020000);
    // (the launcher sends views of 4 GiB or more, and FW_ATTN_VAR bit 10, to the pointer-form instantiation: both forms in ONE kernel
    //  cost hd 128 436 B of scratch)
    auto issue_k = [&](int t, int slot) __attribute__((always_inline)) {
        char* k_lds = smem + (slot & (ARING - 1)) * K_TILE_BYTES;
        if constexpr (BUF) {
            // (the builtin's offset operands are `int`: handed an `unsigned` inside this generic lambda hipcc drops the HOST stub of
            //  every instantiation of the kernel without a diagnostic and the library fails to load)
            const unsigned so = (unsigned)t * (unsigned)k_tile_stride;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (kvalid[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, FW_LDS_PTR(k_lds + (wave + NW * i) * 1024), 16, (int)koff[i], (int)so, 0, 0);
            return;
        }
        const char* kt = Kp + (size_t)t * k_tile_stride;
        const bool last = ragged && t == nt - 1;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            unsigned off = koff[i];
            if (last) {
                const int row = (wave + NW * i) * 4 + (lane >> 4);
                const int over = row - (Lk - 1 - (nt - 1) * KVB);
                if (over > 0) off -= (unsigned)(over * (int)p.ldk) * 2u;
            }
            if (kvalid[i]) FW_GLDS16(kt + off, k_lds + (wave + NW * i) * 1024);
        }
    };
    auto issue_v = [&](int t, int slot) __attribute__((always_inline)) {
        char* v_lds = smem + V_BASE + (slot & (ARING - 1)) * VT_TILE_BYTES;
        if constexpr (BUF) {
            const unsigned so = (unsigned)t * (KVB * 2);
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (lane < VLANES) __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, FW_LDS_PTR(v_lds + (wave + NW * i) * (VROWS * 128)), 16, (int)voff[i], (int)so, 0, 0);
            return;
        }
        const char* vt = Vp + (size_t)t * (KVB * 2);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (lane < VLANES) FW_GLDS16(vt + voff[i], v_lds + (wave + NW * i) * (VROWS * 128));
    };
    int kcoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kcoff[ks] = fi * 256 + (((2 * ks + hi) ^ (fi & 15)) << 4);
    int vcoff[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) vcoff[s2] = V_BASE + fi * 128 + (((2 * s2 + hi) ^ ((fi >> 1) & 7)) << 4);

    f32x16_t o[NS][DB];
    f32x16_t negm[NS];
    float m_run[NS], l_run[NS];
#pragma unroll
    for (int sb = 0; sb < NS; ++sb) {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[sb][d][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[sb][r] = 0.f;
        m_run[sb] = 0.f;
        l_run[sb] = 0.f;
        if (NS == 2) {
#pragma unroll
            for (int d = 0; d < DB; ++d) asm volatile("" : "+a"(o[sb][d]));
        }
    }
    bool bad = false;
    bf16x8_t fr[HF];
    f32x16_t sA[NS], sB[NS];
    uint32_t pw[NS][8];
    f32x4_t lsum = {0.f, 0.f, 0.f, 0.f};                                      // MSUM: every register = this lane's query's row sum
    const uint32_t one2 = (((lane >> 4) ^ (lane >> 2)) & 1) ? 0u : 0x3f803f80u;        // MSUM: this lane's part of the ones pattern
    const u32x4_t one4 = {one2, one2, one2, one2};

    // K fragments of key block `blk` (0/1) of tile t -> fr
    auto load_k_half = [&](int t, int blk) __attribute__((always_inline)) {
        const char* kb = smem + (t & (ARING - 1)) * K_TILE_BYTES + blk * 32 * 256;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fr[ks] = *(const bf16x8_t*)(kb + kcoff[ks]);
    };
    // One half tile: `cur` = shifted scores of half u (consumed), `nxt` receives those of half u+1.
    // NEXT: half u+1 exists;  NEXT2: half u+2 exists;  MASK: last tile (masks the keys beyond Lk when it is ragged).
    // The shift is the row maximum of the first 32 keys and never moves: exp2 of a log2-domain score minus that anchor stays
    // a finite float (and a finite bf16) for spreads up to ~2^96, and floating point keeps the relative precision whatever
    // the magnitude, so there is nothing to rescale.  A half-row sum beyond 2^96 (or a NaN) only raises `bad`; such a wave
    // recomputes its rows with the exact recurrence after the loop.  The loop body therefore has no data-dependent branch.
    // Every K / Vt fragment feeds the MFMAs of all NS sub-blocks (NS = 2 halves the LDS reads per MFMA).
    auto half = [&](f32x16_t (&cur)[NS], f32x16_t (&nxt)[NS], int t, auto slot_tag, auto hf_tag, auto next_tag, auto next2_tag,
                    auto mask_tag) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_tag)::value;        // ring slot of tile t when known at compile time, -1 = t & (ARING - 1)
        constexpr int hf = decltype(hf_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value, NEXT2 = decltype(next2_tag)::value, MASK = decltype(mask_tag)::value;
        if (MASK) {
            if (ragged) {
                const int kbase = t * KVB + hf * 32 + 4 * hi;
#pragma unroll
                for (int sb = 0; sb < NS; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + (r & 3) + 8 * (r >> 2) >= Lk) cur[sb][r] = -1.0e30f;
            }
        }
        const char* vb = smem + (SL >= 0 ? SL : (t & (ARING - 1))) * VT_TILE_BYTES;
        // ---- stage A: QK^T of half u+1 (fr = its K fragments), softmax of half u, Vt fragments of half u into the freed registers
        float ls[NS];
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) ls[sb] = 0.f;
#pragma unroll
        for (int j = 0; j < HF; ++j) {
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                if (NEXT && !AB_NOQK) {
                    if (NS == 2) {
                        if (j == 0) fw_mfma_s_first(nxt[sb], fr[j], qf[sb][j], negm[sb]);
                        else fw_mfma_s_acc(nxt[sb], fr[j], qf[sb][j]);
                    } else {
                        nxt[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j], qf[sb][j], j == 0 ? negm[sb] : nxt[sb], 0, 0, 0);
                    }
                }
                // pairs (2i, 2i+1), i in [8j/HF, 8(j+1)/HF): exp, row sum and bf16 pack ride behind this MFMA
#pragma unroll
                for (int i = (8 * j) / HF; i < (8 * (j + 1)) / HF; ++i) {
                    const float a0 = AB_NOEXP ? cur[sb][2 * i] : __builtin_amdgcn_exp2f(cur[sb][2 * i]);
                    const float a1 = AB_NOEXP ? cur[sb][2 * i + 1] : __builtin_amdgcn_exp2f(cur[sb][2 * i + 1]);
                    if (!MSUM) ls[sb] += a0 + a1;
                    pw[sb][i] = pack_bf16x2(a0, a1);
                }
                if (sb == NS - 1 && !AB_NODS) fr[j] = *(const bf16x8_t*)(vb + (j % DB) * 32 * 128 + vcoff[2 * hf + j / DB]);
                if (NS == 2) __builtin_amdgcn_sched_barrier(0);      // one MFMA and its fillers per scheduling region
            }
        }
        if (PINNED && NS == 1) {
#pragma unroll
            for (int j = 0; j < HF; ++j) {
                if (NEXT) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 // DS read
                __builtin_amdgcn_sched_group_barrier(0x400, 2 * PPJ, 0);           // TRANS (v_exp)
                __builtin_amdgcn_sched_group_barrier(0x002, 3 * PPJ, 0);           // VALU (sum, pack)
            }
            if (PREM) {
                __builtin_amdgcn_sched_group_barrier(0x400, 2 * PREM, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3 * PREM, 0);
            }
        }
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) {
            // (round 3: moving this test out of the loop -- one test of l_run at the end bounds every half sum -- saves 3 VALU per half
            //  on paper; in the ring-unrolled build it tipped the register allocator over the 256-VGPR edge (296 B of scratch, 126
            //  scratch accesses per four tiles), so it stays.)
            if (!MSUM) {
                bad |= !(ls[sb] <= 0x1p96f);
                l_run[sb] += ls[sb];
            }
        }
        // ---- stage B: PV of half u, K fragments of half u+2 behind the MFMAs
        const char* kb = smem + (SL >= 0 ? ((SL + 1) & (ARING - 1)) : ((t + 1) & (ARING - 1))) * K_TILE_BYTES + hf * 32 * 256;      // half u+2 = tile t+1, same block
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            const int ks2 = i / DB, d = i % DB;
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                u32x4_t pv4 = {pw[sb][4 * ks2], pw[sb][4 * ks2 + 1], pw[sb][4 * ks2 + 2], pw[sb][4 * ks2 + 3]};
                bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
                if (AB_NOPV) asm volatile("" :: "v"(pf));
                else if (NS == 2) fw_mfma_o_acc(o[sb][d], fr[i], pf);
                else o[sb][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], pf, o[sb][d], 0, 0, 0);
                if (MSUM && d == DB - 1)           // behind the last PV MFMA that reads this chunk of P^T
                    lsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, one4), pf, lsum, 0, 0, 0);
            }
            if (NEXT2 && !AB_NODS) fr[i] = *(const bf16x8_t*)(kb + kcoff[i]);
            if (NS == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (PINNED && NS == 1) {
#pragma unroll
            for (int i = 0; i < HF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (NEXT2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    // one 64-key tile.  The ring slot of K(t+3) held K(t-1) and that of Vt(t+2) held Vt(t-2), both dead; near the end the
    // requests are clamped to the last tile (they land in slots nobody reads) so the wait count stays a constant.
    using SR = std::integral_constant<int, -1>;
    auto tile = [&](int t, auto slot_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        using NotLast = std::integral_constant<bool, !LAST>;
        half(sA, sB, t, slot_tag, H0{}, T_{}, NotLast{}, last_tag);
        if (!LAST && !AB_NODMA) {
            issue_k(min(t + 3, nt - 1), t + 3);
            issue_v(min(t + 2, nt - 1), t + 2);
        }
        half(sB, sA, t, slot_tag, H1{}, NotLast{}, NotLast{}, last_tag);
        if (!LAST && !AB_NODMA) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fw_await_vm<2 * NP>();
            FW_ABARRIER();
        }
    };

    if ((p.prio == 1 && wave >= NW / 2) || (p.prio == 2 && wave < NW / 2)) __builtin_amdgcn_s_setprio(1);
    // ---- prologue ---------------------------------------------------------------------------------------------------
    issue_k(0, 0);
    issue_k(min(1, nt - 1), 1);
    issue_k(min(2, nt - 1), 2);
    issue_v(0, 0);
    issue_v(min(1, nt - 1), 1);
    fw_await_vm<0>();
    FW_ABARRIER();
    load_k_half(0, 0);
    if (NS == 2) asm volatile("s_nop 7" ::: "memory");      // VALU-written zero splat -> srcC of the first asm MFMA
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int sb = 0; sb < NS; ++sb)
        {
            if (NS == 2) {
                if (ks == 0) fw_mfma_s_first(sA[sb], fr[ks], qf[sb][ks], negm[sb]);
                else fw_mfma_s_acc(sA[sb], fr[ks], qf[sb][ks]);
            } else {
                sA[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks], qf[sb][ks], ks == 0 ? negm[sb] : sA[sb], 0, 0, 0);
            }
        }
    if (NS == 2) fw_mfma_drain();
    load_k_half(0, 1);                        // K fragments of half 1 for stage A of half 0
#pragma unroll
    for (int sb = 0; sb < NS; ++sb) {
        // anchor: exact row maximum of the first 32 keys (keys beyond Lk excluded)
        if (ragged && nt == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (4 * hi + (r & 3) + 8 * (r >> 2) >= Lk) sA[sb][r] = -1.0e30f;
        }
        float mx = fw_max16(sA[sb]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        m_run[sb] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sA[sb][r] -= m_run[sb];
            negm[sb][r] = -m_run[sb];
        }
    }

    // the shift splat was just written by VALU and the first asm MFMA reads it as srcC: that hazard is ours to cover
    if (NS == 2) asm volatile("s_nop 7" ::: "memory");
    if (PAIR) {
        // Two tiles per barrier.  Tile t reads K(t+1) and Vt(t) from LDS, so a pair (t, t+1) needs K(t+1), K(t+2), Vt(t),
        // Vt(t+1) landed at its entry and leaves K(<= t), Vt(< t) dead: K(t+3), K(t+4), Vt(t+2), Vt(t+3) are requested at the
        // entry of the pair (their ring slots hold K(t-1), K(t), Vt(t-2), Vt(t-1)) and awaited in full at its end -- one
        // pair time (~3 us) of prefetch.  The extra barrier keeps the first request for K(4) off slot 0 until every wave
        // has taken its K(0) fragments.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FW_ABARRIER();
        int t = 0;
        for (; t + 2 < nt; t += 2) {
            issue_k(min(t + 3, nt - 1), t + 3);
            issue_k(min(t + 4, nt - 1), t + 4);
            issue_v(min(t + 2, nt - 1), t + 2);
            issue_v(min(t + 3, nt - 1), t + 3);
            half(sA, sB, t, SR{}, H0{}, T_{}, T_{}, F_{});
            half(sB, sA, t, SR{}, H1{}, T_{}, T_{}, F_{});
            half(sA, sB, t + 1, SR{}, H0{}, T_{}, T_{}, F_{});
            half(sB, sA, t + 1, SR{}, H1{}, T_{}, T_{}, F_{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fw_await_vm<0>();
            FW_ABARRIER();
        }
        if (t + 1 < nt) {                     // a non-last tile whose data the last pair already brought in
            half(sA, sB, t, SR{}, H0{}, T_{}, T_{}, F_{});
            half(sB, sA, t, SR{}, H1{}, T_{}, T_{}, F_{});
            ++t;
        }
        tile(t, SR{}, T_{});
    } else if (UNR) {
        int t = 0;
        static_assert(ARING == 4, "the bodies below are unrolled by hand over ring slots 0..3");
#pragma unroll 1
        for (; t + ARING <= nt - 1; t += ARING) {       // t is a multiple of the ring depth here: slots 0, 1, 2, 3
            tile(t, std::integral_constant<int, 0>{}, F_{});
            tile(t + 1, std::integral_constant<int, 1>{}, F_{});
            tile(t + 2, std::integral_constant<int, 2>{}, F_{});
            tile(t + 3, std::integral_constant<int, 3>{}, F_{});
        }
#pragma unroll 1
        for (; t < nt - 1; ++t) tile(t, SR{}, F_{});
        tile(nt - 1, SR{}, T_{});
    } else {
        if (nt > 1) {
            for (int t = 0; t < nt - 1; ++t) tile(t, SR{}, F_{});
        }
        tile(nt - 1, SR{}, T_{});
    }
    if (NS == 2) fw_mfma_drain();
    if (MSUM) {
        // half of the total per half-wave, so that the combine below (l_run + its partner's) and the split-KV record stay as they are
        bad |= !(lsum[0] <= 0x1p96f);
        l_run[0] = 0.5f * lsum[0];
    }

    if (__any(bad)) {
        // ---- exact recomputation of this wave's rows: plain online softmax, fragments straight from global memory.
        // Only reached when the score spread of a row exceeds ~2^96 in the log2 domain; speed does not matter here.
        const uint16_t* Kg = (const uint16_t*)Kp;
        const uint16_t* Vg = (const uint16_t*)Vp;
#pragma unroll 1
        for (int sb = 0; sb < NS; ++sb) {
            f32x16_t oe[DB];
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oe[d][r] = 0.f;
            float me = -3.0e38f, le = 0.f;
            bf16x8_t qe[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qe[ks] = sb == 0 ? qf[0][ks] : qf[NS - 1][ks];
            for (int kb0 = 0; kb0 < Lk; kb0 += 32) {
                f32x16_t sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
                const int krow = min(kb0 + fi, Lk - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(Kg + (size_t)krow * p.ldk + ks * 16 + hi * 8);
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qe[ks], sc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb0 + 4 * hi + (r & 3) + 8 * (r >> 2) >= Lk) sc[r] = -3.0e38f;
                float mx = fw_max16(sc);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])), me);
                const float alpha = __builtin_amdgcn_exp2f(fmaxf(me - mx, -200.f));
                me = mx;
                le *= alpha;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oe[d][r] *= alpha;
                uint32_t pq[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a0 = __builtin_amdgcn_exp2f(fmaxf(sc[2 * i] - mx, -200.f));
                    const float a1 = __builtin_amdgcn_exp2f(fmaxf(sc[2 * i + 1] - mx, -200.f));
                    le += a0 + a1;
                    pq[i] = pack_bf16x2(a0, a1);
                }
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    u32x4_t pv4 = {pq[4 * ks2], pq[4 * ks2 + 1], pq[4 * ks2 + 2], pq[4 * ks2 + 3]};
                    const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv4);
#pragma unroll
                    for (int d = 0; d < DB; ++d) {
                        const bf16x8_t vf = *(const bf16x8_t*)(Vg + (size_t)(d * 32 + fi) * p.lkp + kb0 + ks2 * 16 + hi * 8);
                        oe[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oe[d], 0, 0, 0);
                    }
                }
            }
            if (sb == 0) {
#pragma unroll
                for (int d = 0; d < DB; ++d) o[0][d] = oe[d];
                l_run[0] = le;
                m_run[0] = me;
            } else {
#pragma unroll
                for (int d = 0; d < DB; ++d) o[NS - 1][d] = oe[d];
                l_run[NS - 1] = le;
                m_run[NS - 1] = me;
            }
        }
    }

    if (p.part) {
        // split-KV: un-normalised O^T, row sum and shift of this run; rows beyond Lq are written too (never read back)
#pragma unroll
        for (int sb = 0; sb < NS; ++sb) {
            const float l_tot = l_run[sb] + __shfl_xor(l_run[sb], 32, 64);
            float* dst = p.part + ((((int64_t)bh * p.nqb + qb) * p.nsplit + run) * QB + (wave * NS + sb) * 32 + fi) * (HD + 2);
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4_t v = {o[sb][d][4 * g + 0], o[sb][d][4 * g + 1], o[sb][d][4 * g + 2], o[sb][d][4 * g + 3]};
                    *(f32x4_t*)(dst + d * 32 + 8 * g + 4 * hi) = v;
                }
            if (hi == 0) { dst[HD] = l_tot; dst[HD + 1] = m_run[sb]; }
        }
        return;
    }
    if (NS == 1) {
        // The accumulators are FINAL before the only divergent branch of the kernel (round 6): an MFMA that the compiler sinks into the
        // rows-below-Lq guard would run under a partial EXEC, and that returns garbage (seen in one fp8 instantiation, csrc/attention_fp8.hip).
        // hipcc does not do it to this kernel today; this keeps it from ever doing it.
#pragma unroll
        for (int d = 0; d < DB; ++d) asm volatile("" : "+v"(o[0][d]));
    }
#pragma unroll
    for (int sb = 0; sb < NS; ++sb) {
        const float l_tot = l_run[sb] + __shfl_xor(l_run[sb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q_row = qb * QB + (wave * NS + sb) * 32 + fi;
        if (q_row < p.Lq) {
            uint16_t* dst = Op + (int64_t)q_row * p.ldo;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = d * 32 + 8 * g + 4 * hi;
                    float v0 = o[sb][d][4 * g + 0] * inv, v1 = o[sb][d][4 * g + 1] * inv;
                    float v2 = o[sb][d][4 * g + 2] * inv, v3 = o[sb][d][4 * g + 3] * inv;
                    u32x2_t* ptr = (u32x2_t*)(dst + col);
                    if (p.accumulate) {
                        const u32x2_t old = *ptr;
                        v0 += __uint_as_float(old[0] << 16);
                        v1 += __uint_as_float(old[0] & 0xffff0000u);
                        v2 += __uint_as_float(old[1] << 16);
                        v3 += __uint_as_float(old[1] & 0xffff0000u);
                    }
                    u32x2_t w = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    *ptr = w;
                }
            }
        }
    }
}

// Merge of the split-KV runs: out[row] = sum_c O_c 2^(m_c - M) / sum_c l_c 2^(m_c - M), M = max_c m_c (all in the log2 domain the
// kernels work in).  One wave per (batch, head, q-block, row); lane = 2 or 4 consecutive features.
template <int HD>
__global__ __launch_bounds__(256) void attention_combine_kernel(AttnArgs p) {
    const int lane = threadIdx.x & 63;
    const int64_t witem = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t rows_total = (int64_t)p.batch * p.heads * p.nqb * QB;
    if (witem >= rows_total) return;
    const int row_in = (int)(witem % QB);
    const int64_t blk = witem / QB;                     // (bh * nqb + qb)
    const int qb = (int)(blk % p.nqb);
    const int bh = (int)(blk / p.nqb);
    const int q_row = qb * QB + row_in;
    if (q_row >= p.Lq) return;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const float* base = p.part + (blk * p.nsplit * QB + row_in) * (HD + 2);
    const int64_t run_stride = (int64_t)QB * (HD + 2);
    float M = -3.0e38f;
    for (int c = 0; c < p.nsplit; ++c) M = fmaxf(M, base[c * run_stride + HD + 1]);
    constexpr int PER = HD / 64 + (HD % 64 ? 1 : 0);    // features per lane: 2 (hd 128), 2 (hd 96: 48 lanes active), 1 (hd 64)
    constexpr int W = HD == 64 ? 1 : 2;
    const int f0 = lane * W;
    if (f0 >= HD) return;
    float acc[W], L = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.f;
    for (int c = 0; c < p.nsplit; ++c) {
        const float* src = base + c * run_stride;
        const float wgt = __builtin_amdgcn_exp2f(fmaxf(src[HD + 1] - M, -200.f));
        L += src[HD] * wgt;
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] += src[f0 + j] * wgt;
    }
    (void)PER;
    const float inv = 1.0f / L;
    uint16_t* dst = p.O + (int64_t)b * p.bso + (int64_t)h * HD + (int64_t)q_row * p.ldo + f0;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        float v = acc[j] * inv;
        if (p.accumulate) v += bf16_bits_to_f32(dst[j]);
        dst[j] = f32_to_bf16_bits(v);
    }
}

// V[b][Lk][heads*hd] -> Vt[b][h][d][Lk_pad]; position p inside each 32-key block holds key swap_bits_2_3(p).
// One work-group transposes a 64-key x 64-channel tile through LDS.
__global__ __launch_bounds__(256) void v_transpose_kernel(const uint16_t* __restrict__ V, int64_t ldv, int64_t bsv,
                                                          uint16_t* __restrict__ Vt, int64_t lkp,
                                                          int heads, int hd, int Lk) {
    __shared__ uint16_t tile[64][66];
    const int k0 = blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;            // channel offset inside heads*hd
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int width = heads * hd;
    // load: 64 keys x 64 channels, 8 channels (16 B) per thread-iteration
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;        // 0..511
        const int kr = idx >> 3, cc = (idx & 7) * 8;
        const int key = k0 + kr;
        u32x4_t v = {0, 0, 0, 0};
        if (key < Lk && c0 + cc < width) v = *(const u32x4_t*)(V + (int64_t)b * bsv + (int64_t)key * ldv + c0 + cc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[kr][cc + 2 * j] = (uint16_t)(v[j] & 0xffffu);
            tile[kr][cc + 2 * j + 1] = (uint16_t)(v[j] >> 16);
        }
    }
    __syncthreads();
    // store: channel row ch, 64 positions; each thread writes 8 consecutive positions (16 B)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;
        const int ch = idx >> 3, p0 = (idx & 7) * 8;
        const int chan = c0 + ch;
        if (chan >= width) continue;
        const int h = chan / hd, d = chan - h * hd;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pa = p0 + 2 * j, pb = pa + 1;
            const int ka = (pa & ~12) | ((pa & 4) << 1) | ((pa & 8) >> 1);
            const int kb = (pb & ~12) | ((pb & 4) << 1) | ((pb & 8) >> 1);
            w[j] = (uint32_t)tile[ka][ch] | ((uint32_t)tile[kb][ch] << 16);
        }
        u32x4_t o4 = {w[0], w[1], w[2], w[3]};
        *(u32x4_t*)(Vt + (((int64_t)b * heads + h) * hd + d) * lkp + k0 + p0) = o4;
    }
}

}  // namespace

// Split-KV plan for the tail q-block.  It depends on (heads, Lq, Lk) ONLY -- never on the batch: the merged CFG pass runs the
// same attention with batch 2 (FusionEngine.joint_forward_pair) and must merge a tail row's key runs in the same order as the
// two separate forwards do, or the pair is no longer bit-identical to them (round-2 advisor finding: target = 256 / (heads *
// batch) gave 16 runs at batch 1 and 8 at batch 2 for the VGGT global attention at L2 = 32865).  The criterion is the
// per-sample grid: the tail work-groups of ONE sample would open a new round of the 256 CUs.
// Requests by descriptor carry byte offsets in 32 bits (lane offset + scalar tile offset, tiles up to 4 past the last one).
static bool attn_view_needs_pointers(int Lk, int64_t ldk, int head_dim, int64_t Lk_pad) {
    const size_t lim = 0xffffffffull;
    return ((size_t)Lk + 5 * KVB) * (size_t)ldk * 2 >= lim || ((size_t)head_dim * (size_t)Lk_pad + 5 * KVB) * 2 >= lim;
}

static bool attn_split_plan(int heads, int Lq, int Lk, int* nsplit, int* tps) {
    const int tail = Lq % QB, nqb_full = Lq / QB;
    const int nt = (Lk + KVB - 1) / KVB;
    if (tail == 0 || nqb_full == 0 || nt < 16 || heads > 128) return false;
    const int64_t w_full = (int64_t)nqb_full * heads, w_all = w_full + heads;
    if ((w_full + 255) / 256 >= (w_all + 255) / 256) return false;             // the tail work-groups fit the last round anyway
    const int target = max(2, min(16, 256 / heads));
    *tps = (nt + target - 1) / target;
    *nsplit = (nt + *tps - 1) / *tps;
    return true;
}

// Bytes of workspace with which fw_attention_bf16 takes the split-KV route for the tail q-block; 0 = it would not use one.
extern "C" int64_t fw_attention_workspace_bytes(int batch, int heads, int head_dim, int Lq, int Lk) {
    if (batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0) return 0;
    int nsplit = 0, tps = 0;
    if (!attn_split_plan(heads, Lq, Lk, &nsplit, &tps)) return 0;
    return (int64_t)batch * heads * nsplit * QB * (head_dim + 2) * (int64_t)sizeof(float);
}

extern "C" int fw_attention_bf16(const uint16_t* Q, int64_t ldq, int64_t bsq,
                                 const uint16_t* K, int64_t ldk, int64_t bsk,
                                 const uint16_t* Vt, int64_t Lk_pad,
                                 uint16_t* O, int64_t ldo, int64_t bso,
                                 int batch, int heads, int head_dim, int Lq, int Lk,
                                 float scale, int flags, void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch <= 0 || heads <= 0 || Lq <= 0) return 0;
    if (Lk <= 0) { fw_set_error("fw_attention_bf16: Lk must be > 0"); return FW_E_BADARG; }
    if (head_dim != 64 && head_dim != 96 && head_dim != 128) { fw_set_error("fw_attention_bf16: head_dim must be 64, 96 or 128"); return FW_E_UNSUPPORTED; }
    if ((ldq % 8) || (ldk % 8) || (ldo % 4) || (bsq % 8) || (bsk % 8) || (bso % 4) || (Lk_pad % 64) || Lk_pad < Lk ||
        (((uintptr_t)Q) & 15) || (((uintptr_t)K) & 15) || (((uintptr_t)Vt) & 15) || (((uintptr_t)O) & 7)) {
        fw_set_error("fw_attention_bf16: alignment contract violated (16-B Q/K/Vt, 8-B O, Lk_pad % 64 == 0)"); return FW_E_BADARG; }
    AttnArgs p;
    p.Q = Q; p.ldq = ldq; p.bsq = bsq; p.K = K; p.ldk = ldk; p.bsk = bsk; p.Vt = Vt; p.lkp = Lk_pad;
    p.O = O; p.ldo = ldo; p.bso = bso; p.batch = batch; p.heads = heads; p.Lq = Lq; p.Lk = Lk;
    const bool prescaled = (flags & FW_ATTN_Q_PRESCALED) != 0;
    p.scale_log2 = prescaled ? 1.0f : scale * 1.4426950408889634f; p.accumulate = (flags & FW_ATTN_ACCUMULATE) ? 1 : 0;
    p.nqb = (Lq + QB - 1) / QB;
    p.part = nullptr; p.nsplit = 1; p.tiles_per_split = 0;
    p.prio = (fw_get_option(FW_OPT_ATTN_VAR) >> 8) & 3;
    p.nofast = (fw_get_option(FW_OPT_ATTN_VAR) >> 10) & 1;      // A/B: 1024 = tile requests in the round-3 pointer form
    const int64_t nwg = (int64_t)p.nqb * heads * batch;
    if (nwg > 0x7fffffff) { fw_set_error("fw_attention_bf16: grid too large"); return FW_E_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    // Tail q-block (Lq % 256 != 0) that would cost the grid a whole extra round of the 256 CUs (one work-group per CU): its rows
    // go through split-KV instead -- the full q-blocks in one launch (whole rounds), then the tail rows with the keys cut into
    // runs (heads * batch * nsplit short work-groups side by side), then the merge.  L2 = 32865 = 128 x 256 + 97 query rows:
    // bicross direction 2 (12 heads) 1548 work-groups = 6.05 rounds -> 6 + 1/16; VGGT global (16 heads) 2064 = 8.06 -> 8 + 1/16.
    if (workspace != nullptr && prescaled) {
        int nsplit = 0, tps = 0;
        const int64_t need = fw_attention_workspace_bytes(batch, heads, head_dim, Lq, Lk);
        if (need > 0 && workspace_bytes >= need && attn_split_plan(heads, Lq, Lk, &nsplit, &tps)) {
            const int Lq_main = (Lq / QB) * QB;
            int rc = fw_attention_bf16(Q, ldq, bsq, K, ldk, bsk, Vt, Lk_pad, O, ldo, bso, batch, heads, head_dim, Lq_main, Lk,
                                       scale, flags, nullptr, 0, stream);
            if (rc) return rc;
            AttnArgs t = p;
            t.Q = Q + (int64_t)Lq_main * ldq; t.O = O + (int64_t)Lq_main * ldo;
            t.Lq = Lq - Lq_main; t.nqb = 1;
            t.part = (float*)workspace; t.nsplit = nsplit; t.tiles_per_split = tps;
            const unsigned g = (unsigned)(heads * batch * nsplit);
            if (head_dim == 128) hipLaunchKernelGGL((attention_sp_kernel<128, 1>), dim3(g), dim3(512), 0, st, t);
            else if (head_dim == 96) hipLaunchKernelGGL((attention_sp_kernel<96, 1>), dim3(g), dim3(512), 0, st, t);
            else hipLaunchKernelGGL((attention_sp_kernel<64, 1>), dim3(g), dim3(512), 0, st, t);
            const unsigned cg = (unsigned)((int64_t)batch * heads * QB + 3) / 4;
            if (head_dim == 128) hipLaunchKernelGGL(attention_combine_kernel<128>, dim3(cg), dim3(256), 0, st, t);
            else if (head_dim == 96) hipLaunchKernelGGL(attention_combine_kernel<96>, dim3(cg), dim3(256), 0, st, t);
            else hipLaunchKernelGGL(attention_combine_kernel<64>, dim3(cg), dim3(256), 0, st, t);
            return (int)hipGetLastError();
        }
    }
    // round 6: the single-stream kernels take their row sums on the matrix pipe (attention_sp_kernel, VAR bit 9); FW_ATTN_VAR bit 11
    // (2048) selects the round-5 choice for the A/B -- fp32 sums on the vector pipe, hd 96 on the two-segment ping-pong kernel
    const bool vsum = ((fw_get_option(FW_OPT_ATTN_VAR) >> 11) & 1) != 0;
    int var = fw_get_option(FW_OPT_ATTN_VAR) & 255;     // 0 = first kernel (generic); 64.. = two-segment ping-pong; 128.. = single stream
    // 192 = per-head-dim choice among the pre-scaled kernels (microbench, profiles/r01/attention_sp_ablation.txt): the
    // single-stream kernel for hd 128 and hd 64, the two-segment ping-pong for hd 96
    // round 3: hd 128 / hd 64 on the ring-unrolled form of the single-stream kernel (193): -2.4 % / -4.6 % on one box
    // (profiles/r03/microbench_attention_unrolled.txt); hd 96 stays on the ping-pong kernel (its unrolled form spills)
    // (hd 64: the unrolled form WITHOUT the sched_group_barrier pins, 196: no scratch, 1.3 % faster than the pinned one)
    // round 6: with the row sums off the vector pipe the unrolled single-stream kernel fits at hd 96 too (186 registers, no scratch)
    // and beats the ping-pong kernel there by 8 % (profiles/r06/attn_ab_msum_call25.txt); hd 64 is faster pinned again
    if (var == 192) var = vsum ? (head_dim == 96 ? 64 : (head_dim == 64 ? 196 : 193)) : 193;
    if ((var & 64) && var >= 128 && head_dim == 96 && (vsum || (var & 2))) var &= ~64;      // (the vector-sum and four-wave forms have no unrolled hd-96 instantiation)
    if (prescaled && var >= 128) {
        // single-stream software pipeline on half tiles, pinned issue order; bit 1: one 64-row wave per SIMD (4 waves)
#define FW_ATTN_SP(HDV, V) \
    hipLaunchKernelGGL((attention_sp_kernel<HDV, V>), dim3((unsigned)nwg), dim3(((V) & 2) ? 256 : 512), 0, st, p)
#define FW_ATTN_SP_HD(V) \
    do { if (head_dim == 128) FW_ATTN_SP(128, V); else if (head_dim == 96) FW_ATTN_SP(96, V); else FW_ATTN_SP(64, V); } while (0)
#define FW_ATTN_SP_HD_UNR(V) do { if (head_dim == 128) FW_ATTN_SP(128, V); else FW_ATTN_SP(64, V); } while (0)
        if (var & 64) {                                                                         // tile loop unrolled by the ring depth
            // (requests by descriptor need every byte offset of the (batch, head) view in 32 bits; else, and for the A/B, pointer form)
            const bool ptr_form = p.nofast || attn_view_needs_pointers(Lk, ldk, head_dim, Lk_pad);
            if (var & 2) FW_ATTN_SP_HD_UNR(67);                // four waves of 64 rows (experiment arm; fp32 sums on the vector pipe)
            else if (!vsum && ptr_form) FW_ATTN_SP_HD(321 + 512);
            else if (!vsum && (var & 4)) FW_ATTN_SP_HD(64 + 512);
            else if (!vsum) FW_ATTN_SP_HD(65 + 512);
            else if (ptr_form) { if (head_dim == 128) FW_ATTN_SP(128, 321); else FW_ATTN_SP(64, 320); }
            else if (var & 4) FW_ATTN_SP_HD_UNR(64);          // 196: without the sched_group_barrier pins
            else FW_ATTN_SP_HD_UNR(65);
        }
        else if (var & 2) FW_ATTN_SP_HD(3); else FW_ATTN_SP_HD(1);
        return (int)hipGetLastError();
    }
    if (prescaled && var >= 64) {
        // two-segment ping-pong on log2-domain scores; bit 1: TIMING build (tools/attn_timeline.py)
#define FW_ATTN_PP3(HDV, V) hipLaunchKernelGGL((attention_pp3_kernel<HDV, V>), dim3((unsigned)nwg), dim3(512), 0, st, p)
        const bool ptr_form = p.nofast || attn_view_needs_pointers(Lk, ldk, head_dim, Lk_pad);
        if (ptr_form) { if (head_dim == 128) FW_ATTN_PP3(128, 8); else if (head_dim == 96) FW_ATTN_PP3(96, 8); else FW_ATTN_PP3(64, 8); }
        else if ((var & 3) == 2) { if (head_dim == 128) FW_ATTN_PP3(128, 2); else if (head_dim == 96) FW_ATTN_PP3(96, 2); else FW_ATTN_PP3(64, 2); }
        else { if (head_dim == 128) FW_ATTN_PP3(128, 0); else if (head_dim == 96) FW_ATTN_PP3(96, 0); else FW_ATTN_PP3(64, 0); }
        return (int)hipGetLastError();
    }
    // q not pre-scaled (callers that cannot fold scale * log2(e) into q), or FW_ATTN_VAR = 0: the generic first kernel
    if (head_dim == 128) hipLaunchKernelGGL(attention_kernel<128>, dim3((unsigned)nwg), dim3(512), 0, st, p);
    else if (head_dim == 96) hipLaunchKernelGGL(attention_kernel<96>, dim3((unsigned)nwg), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(attention_kernel<64>, dim3((unsigned)nwg), dim3(512), 0, st, p);
    return (int)hipGetLastError();
}

extern "C" int fw_v_transpose(const uint16_t* V, int64_t ldv, int64_t bsv, uint16_t* Vt, int64_t Lk_pad,
                              int batch, int heads, int head_dim, int Lk, void* stream) {
    if (batch <= 0 || heads <= 0 || Lk <= 0) return 0;
    if ((ldv % 8) || (bsv % 8) || (Lk_pad % 64) || Lk_pad < Lk || (head_dim % 8) || (((uintptr_t)V) & 15) || (((uintptr_t)Vt) & 15)) {
        fw_set_error("fw_v_transpose: alignment contract violated"); return FW_E_BADARG; }
    const int width = heads * head_dim;
    dim3 grid((unsigned)(Lk_pad / 64), (unsigned)((width + 63) / 64), (unsigned)batch);
    hipLaunchKernelGGL(v_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, V, ldv, bsv, Vt, Lk_pad, heads, head_dim, Lk);
    return (int)hipGetLastError();
}

// Measurement hook (tools/attn_timeline.py): copies the segment timestamps written by the TIMING build of the ping-pong kernel.
extern "C" int fw_debug_attention_timestamps(unsigned long long* host_out, int n) {
    if (n <= 0 || n > 64) { fw_set_error("fw_debug_attention_timestamps: n must be in 1..64"); return FW_E_BADARG; }
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_ts), sizeof(unsigned long long) * n);
}

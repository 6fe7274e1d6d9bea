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
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
    if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_bf16: grid too large"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int fw_gemv_f32(const float* x, const float* W, int64_t ldw, const float* bias, float* out,
                           int N, int K, int act_in_silu, int act_out, void* stream) {
    if (N <= 0 || K <= 0) { fw_set_error("fw_gemv_f32: bad shape"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       x, W, ldw, bias, out, N, K, act_in_silu, act_out);
    return (int)hipGetLastError();
}

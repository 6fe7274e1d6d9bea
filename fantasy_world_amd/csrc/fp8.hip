// fp8 linear: the reference's only fp8 definition, AutoWrappedLinear.fp8_linear
// (FantasyWorld/diffsynth_wan22/vram_management/layers.py:115-151):
//     scale_a[m] = max(max_k |x[m][k]| / 448, 1)                 per-row activation scale, never below 1
//     xq = e4m3(x / (scale_a + 1e-8)),  wq = e4m3(w)              OCP e4m3fn (gfx950's native fp8), weights cast raw
//     y  = (xq wq^T) * scale_a + bf16(bias)   -> out dtype        torch._scaled_mm(xq, wq^T, scale_a, 1, bias, out_dtype)
// Two kernels: a row quantiser (also used with scale = 1 to cast the weights once at pack time) and a GEMM on
// v_mfma_f32_32x32x16_fp8_fp8 with fp32 accumulation and the scale / bias epilogue.  The GEMM is a plain double-buffered
// 128x128x64 tile (4 waves, register prefetch, padded LDS rows): this path is a semantics-complete option, not the tuned bf16 one.
#include "fw_common.h"

namespace {

// The reference divides in fp32 and casts with round-to-nearest-even; a reciprocal-multiply division would flip values that sit on
// a rounding boundary of the 3-bit mantissa, so this file is compiled WITHOUT -ffast-math (csrc/build.sh): `/` is IEEE here.
__device__ __forceinline__ float fp8_prescale(float v, float denom) { return v / denom; }

// xq[m][k] = e4m3(x[m][k] / (scale[m] + 1e-8)), scale[m] = fixed_scale > 0 ? fixed_scale : max(amax_m / 448, 1).  One wave per row.
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                             int64_t ldq, float* __restrict__ scale, int M, int K, int raw) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint16_t* src = x + (int64_t)row * ldx;
    float denom = 1.0f;
    if (!raw) {
        float amax = 0.f;
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

constexpr int FBM = 128, FBN = 128, FBK = 64;     // tile; FBK in fp8 elements = bytes
constexpr int FROW = 72;                          // padded LDS row (bytes): 18 dwords -> conflict-free 8-byte fragment reads

struct Fp8Args {
    const uint8_t* A; int64_t lda; const uint8_t* W; int64_t ldw; const float* scale; const float* bias;
    void* C; int64_t ldc; int out_dtype; int M, N, K, tiles_n;
};

__global__ __launch_bounds__(256) void gemm_fp8_kernel(Fp8Args p) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][2][FBM * FROW];      // [stage][A | W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
    const int m0 = tm * FBM, n0 = tn * FBN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fi = lane & 31, hi = lane >> 5;

    // loader: thread -> (row, 16-byte chunk) twice per operand
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

    // epilogue: D[m][n] with n = lane % 32 (column of the W-side operand), m = (r % 4) + 8 (r / 4) + 4 (lane / 32)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn + 32 * b + fi;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= p.M) continue;
                const float y = acc[a][b][r] * p.scale[m] + bv;
                if (p.out_dtype == FW_DT_F32) ((float*)p.C)[(int64_t)m * p.ldc + n] = y;
                else ((uint16_t*)p.C)[(int64_t)m * p.ldc + n] = f32_to_bf16_bits(y);
            }
        }
}

}  // namespace

extern "C" int fw_fp8_quant_rows(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int M, int K, int raw,
                                 void* stream) {
    if (!x || !q || (!raw && !scale) || K <= 0 || (K % 8) || (ldx % 8) || (ldq % 8) || (((uintptr_t)x) & 15) || (((uintptr_t)q) & 7)) {
        fw_set_error("fw_fp8_quant_rows: K, ldx, ldq must be multiples of 8, x 16-byte and q 8-byte aligned"); return FW_E_BADARG; }
    if (M <= 0) return 0;
    hipLaunchKernelGGL(fp8_quant_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, q, ldq, scale, M, K, raw);
    return (int)hipGetLastError();
}

extern "C" int fw_gemm_fp8(const uint8_t* A, int64_t lda, const uint8_t* W, int64_t ldw, const float* scale_a, const float* bias,
                           void* C, int64_t ldc, int out_dtype, int M, int N, int K, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A || !W || !scale_a || !C || K <= 0 || (K % FBK) || (lda % 16) || (ldw % 16) || (((uintptr_t)A) & 15) || (((uintptr_t)W) & 15)) {
        fw_set_error("fw_gemm_fp8: K must be a positive multiple of 64; A/W 16-byte aligned with lda/ldw % 16 == 0"); return FW_E_BADARG; }
    if (out_dtype != FW_DT_BF16 && out_dtype != FW_DT_F32) { fw_set_error("fw_gemm_fp8: bad out_dtype"); return FW_E_BADARG; }
    Fp8Args p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.scale = scale_a; p.bias = bias; p.C = C; p.ldc = ldc; p.out_dtype = out_dtype;
    p.M = M; p.N = N; p.K = K; p.tiles_n = (N + FBN - 1) / FBN;
    const int64_t nwg = (int64_t)((M + FBM - 1) / FBM) * p.tiles_n;
    if (nwg > 0x7fffffff) { fw_set_error("fw_gemm_fp8: grid too large"); return FW_E_BADARG; }
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

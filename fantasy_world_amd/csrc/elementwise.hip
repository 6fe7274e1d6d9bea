// HBM-bound glue kernels of the denoising forward (gfx950): LayerNorm+modulate, q/k norm + rotary, sinusoidal
// embedding, patchify / unpatchify, VGGT token assembly, fp32->bf16 cast.  All loads/stores are 8-16 B per lane.
#include "fw_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// LayerNorm (+affine) (+ (1+scale)*y + shift), one 256-thread work-group per row, row kept in registers.
// ------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 8;   // float4 per thread -> C <= 8192

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <bool XF32>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const void* __restrict__ xin, int64_t ldx,
                                                            uint16_t* __restrict__ y, int64_t ldy, int C,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int nv = C >> 2;
    f32x4_t v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < nv) {
            if (XF32) {
                v[i] = *(const f32x4_t*)((const float*)xin + (int64_t)row * ldx + idx * 4);
            } else {
                const u32x2_t raw = *(const u32x2_t*)((const uint16_t*)xin + (int64_t)row * ldx + idx * 4);
                v[i][0] = __uint_as_float(raw[0] << 16); v[i][1] = __uint_as_float(raw[0] & 0xffff0000u);
                v[i][2] = __uint_as_float(raw[1] << 16); v[i][3] = __uint_as_float(raw[1] & 0xffff0000u);
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = block_sum_256(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum_256(q, red) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < nv) {
            const int c0 = idx * 4;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = (v[i][j] - mean) * rstd;
                if (w) t = t * w[c0 + j] + (b ? b[c0 + j] : 0.f);
                if (scale) t = t * (1.0f + scale[c0 + j]);
                if (shift) t += shift[c0 + j];
                o[j] = t;
            }
            u32x2_t out = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *(u32x2_t*)(y + (int64_t)row * ldy + c0) = out;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Fast path: ONE WAVE per row (4 rows per 256-thread group), row held in registers, statistics by wave shuffles:
// no LDS, no barriers, VPL independent 16-B loads in flight per lane.  C = 4 * 64 * k, k <= LNW_MAXV (C <= 5120).
// ------------------------------------------------------------------------------------------------------------
constexpr int LNW_MAXV = 20;

// The normalise / modulate arithmetic of BOTH LayerNorm wave kernels, spelled out operation by operation (no contraction, no
// reassociation): u = (v - mean) * rstd;  [u = u * w (+ b)];  u = fma(u, scale, u) = u * (1 + scale) in one rounding;  u = u + shift.
// A sequence shard (few rows per rank: one-wave-per-row kernel) and the unsharded forward (LDS-staged kernel) must agree bit for bit.
__device__ __forceinline__ f32x4_t ln_finish(f32x4_t v, float mean, float rstd, bool has_w, f32x4_t w4, bool has_b, f32x4_t b4,
                                             bool has_scale, f32x4_t sc, bool has_shift, f32x4_t sh) {
    // Every fusion is spelled out (explicit fma) and every place where the back end could still contract or reassociate under
    // -ffast-math (it honours the function-level unsafe-fp-math attribute, not the per-statement pragmas) is fenced by an empty asm:
    // the two LayerNorm kernels must emit the same arithmetic whatever their surrounding code looks like.
    f32x4_t t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float u = v[j] - mean;
        asm volatile("" : "+v"(u));
        u = u * rstd;
        asm volatile("" : "+v"(u));
        if (has_w) u = has_b ? __builtin_fmaf(u, w4[j], b4[j]) : u * w4[j];
        if (has_scale) u = __builtin_fmaf(u, sc[j], u);
        if (has_shift) { asm volatile("" : "+v"(u)); u = u + sh[j]; }
        t[j] = u;
    }
    return t;
}

// Row statistics of BOTH LayerNorm wave kernels, with the summation order pinned (no reassociation under -ffast-math): the LDS-staged
// kernel (unsharded forward) and the one-wave-per-row kernel (a sequence shard's few rows) must produce the same mean / rstd bits.
// (Round 3: with VPL = 4 the compiler associated the two kernels' sums differently -- caught by the sharded-vs-unsharded bit test.)
template <int VPL>
__device__ __forceinline__ void ln_stats(const f32x4_t (&v)[VPL], int C, float eps, float& mean, float& rstd) {
    // pinned order: chunk sums (a + b) + (c + d) accumulated left to right, squares by explicit fma, 1 / C as one reciprocal used
    // by both statistics; the asm fences keep the back end from re-associating / contracting across statements
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i][0] + v[i][1], b = v[i][2] + v[i][3];
        asm volatile("" : "+v"(a), "+v"(b));
        float c = a + b;
        asm volatile("" : "+v"(c));
        s = s + c;
        asm volatile("" : "+v"(s));
    }
    float inv_c = 1.0f / (float)C;
    asm volatile("" : "+v"(inv_c));
    mean = wave_sum(s) * inv_c;
    asm volatile("" : "+v"(mean));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = v[i][j] - mean;
            asm volatile("" : "+v"(d));
            q = __builtin_fmaf(d, d, q);
        }
    rstd = rsqrtf(__builtin_fmaf(wave_sum(q), inv_c, eps));
}

template <bool XF32, int VPL>
__global__ __launch_bounds__(256) void layernorm_mod_wave_kernel(const void* __restrict__ xin, int64_t ldx,
                                                                 uint16_t* __restrict__ y, int64_t ldy, int rows, int C,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float eps, uint16_t* __restrict__ ylo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4_t v[VPL];
    if (XF32) {
        const float* xr = (const float*)xin + (int64_t)row * ldx;
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4_t*)(xr + (lane + 64 * i) * 4);
    } else {
        const uint16_t* xr = (const uint16_t*)xin + (int64_t)row * ldx;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const u32x2_t raw = *(const u32x2_t*)(xr + (lane + 64 * i) * 4);
            v[i][0] = __uint_as_float(raw[0] << 16); v[i][1] = __uint_as_float(raw[0] & 0xffff0000u);
            v[i][2] = __uint_as_float(raw[1] << 16); v[i][3] = __uint_as_float(raw[1] & 0xffff0000u);
        }
    }
    float mean, rstd;
    ln_stats<VPL>(v, C, eps, mean, rstd);
    uint16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c0 = (lane + 64 * i) * 4;
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t t = ln_finish(v[i], mean, rstd, w != nullptr, w ? *(const f32x4_t*)(w + c0) : zero,
                                    w != nullptr && b != nullptr, (w && b) ? *(const f32x4_t*)(b + c0) : zero,
                                    scale != nullptr, scale ? *(const f32x4_t*)(scale + c0) : zero,
                                    shift != nullptr, shift ? *(const f32x4_t*)(shift + c0) : zero);
        u32x2_t out = {pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
        *(u32x2_t*)(yr + c0) = out;
        if (ylo) {      // fw_layernorm_mod_split: the part of t the bf16 rounding dropped, as a second bf16 (hi + lo carries ~16 bits)
            const float r0 = t[0] - __uint_as_float(out[0] << 16), r1 = t[1] - __uint_as_float(out[0] & 0xffff0000u);
            const float r2 = t[2] - __uint_as_float(out[1] << 16), r3 = t[3] - __uint_as_float(out[1] & 0xffff0000u);
            u32x2_t lo = {pack_bf16x2(r0, r1), pack_bf16x2(r2, r3)};
            *(u32x2_t*)(ylo + (int64_t)row * ldy + c0) = lo;
        }
    }
}

// Parameter vectors staged ONCE per work-group in LDS, work-groups walking the rows.  The hot case is the AdaLN-modulated LayerNorm
// of the fp32 DiT stream without affine weights (wan_video_dit.py:301-310; 290 launches per step): the one-wave-per-row kernel above
// re-reads 40 KiB of scale / shift through the texture path for every 20 KiB row it normalises (60 vector loads per row instead of
// 20) and measures 4.7 TB/s where torch's plain fp32 -> bf16 cast reaches 5.85 (tools/probes/stream_bw.py).  Round 3 (kernel trace
// of the bench at HEAD): the same disease on the AFFINE forms -- DiT norm3 (w, b; 426 us per launch against 189 us for the modulated
// form of the same rows) and the VGGT norm1 / norm2 (w, b [, scale, shift] on 4 KiB rows: 16 KiB of parameters per row, 1.95 TB/s)
// -- so AFF / MOD are template flags and every combination stages what it uses.  Same ln_finish arithmetic as the wave kernel:
// bit-identical results.
template <int VPL, bool AFF, bool MOD>
__global__ __launch_bounds__(256) void layernorm_lds_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y,
                                                            int64_t ldy, int rows, int C, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float eps) {
    __shared__ f32x4_t smW[AFF ? VPL * 64 : 1], smBi[AFF ? VPL * 64 : 1], smA[MOD ? VPL * 64 : 1], smB[MOD ? VPL * 64 : 1];
    for (int i = threadIdx.x; i < VPL * 64; i += 256) {
        if (AFF) { smW[i] = *(const f32x4_t*)(w + i * 4); smBi[i] = *(const f32x4_t*)(b + i * 4); }
        if (MOD) { smA[i] = *(const f32x4_t*)(scale + i * 4); smB[i] = *(const f32x4_t*)(shift + i * 4); }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        f32x4_t v[VPL];
        const float* xr = x + (int64_t)row * ldx;
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4_t*)(xr + (lane + 64 * i) * 4);
        float mean, rstd;
        ln_stats<VPL>(v, C, eps, mean, rstd);
        uint16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t t = ln_finish(v[i], mean, rstd, AFF, AFF ? smW[c] : zero, AFF, AFF ? smBi[c] : zero,
                                        MOD, MOD ? smA[c] : zero, MOD, MOD ? smB[c] : zero);
            u32x2_t out = {pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
            *(u32x2_t*)(yr + c * 4) = out;
        }
    }
}

template <bool XF32>
static bool launch_ln_wave(hipStream_t st, const void* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int C,
                           const float* w, const float* b, const float* scale, const float* shift, float eps, uint16_t* ylo = nullptr) {
    if (XF32 && rows >= 4096 && (C == 5120 || C == 1024) && !ylo) {
        // parameters staged in LDS (20 KiB per vector at C = 5120: 4 work-groups per CU; 4 KiB at C = 1024: 8 per CU), rows walked
        const bool aff = w && b, mod = scale && shift;
        const bool pure = (aff || (!w && !b)) && (mod || (!scale && !shift));       // no half-specified pairs on this path
        const int wgs = min((rows + 3) / 4, 256 * (C == 5120 ? 4 : 8));
#define FW_LN_LDS(V, A, M) do { hipLaunchKernelGGL((layernorm_lds_kernel<V, A, M>), dim3(wgs), dim3(256), 0, st, (const float*)x, ldx, y, ldy, rows, C, w, b, scale, shift, eps); return true; } while (0)
        if (pure && (aff || mod)) {
            if (C == 5120) { if (aff && mod) FW_LN_LDS(20, true, true); else if (aff) FW_LN_LDS(20, true, false); else FW_LN_LDS(20, false, true); }
            else { if (aff && mod) FW_LN_LDS(4, true, true); else if (aff) FW_LN_LDS(4, true, false); else FW_LN_LDS(4, false, true); }
        }
#undef FW_LN_LDS
    }
    const dim3 grid((rows + 3) / 4), block(256);
#define FW_LN_CASE(V) case V: hipLaunchKernelGGL((layernorm_mod_wave_kernel<XF32, V>), grid, block, 0, st, x, ldx, y, ldy, rows, C, w, b, scale, shift, eps, ylo); return true;
    switch (C / 256) {
        FW_LN_CASE(4) FW_LN_CASE(5) FW_LN_CASE(8) FW_LN_CASE(20)
        default: return false;
    }
#undef FW_LN_CASE
}

// ------------------------------------------------------------------------------------------------------------
// q/k normalisation + rotary, in place on a bf16 [rows][heads*hd] slice.  One work-group per row; each thread
// owns up to QK_MAXC chunks of 8 consecutive channels; the normalised row is parked in LDS (fp32) so the rotary
// partner (i^1 for interleaved pairs, i +- hd/4 for the 2-D rotate-half form) can be fetched by any thread.
// ------------------------------------------------------------------------------------------------------------
// a * c -/+ b * s of the rotary embedding with the contraction SPELLED OUT (one product rounded, the other fused): left to
// -ffp-contract=fast the compiler picked the fused operand per INSTANTIATION, and the e4m3-writing form of the DiT q/k pass differed
// from the bf16 form in the last fp32 bit of some elements (round 6: fw_qk_prep_fp8 must return the bits of cast(fw_qk_prep(.)))
__device__ __forceinline__ float qk_rot_sub(float a, float c, float b, float s) { return __builtin_fmaf(a, c, -(b * s)); }
__device__ __forceinline__ float qk_rot_add(float a, float c, float b, float s) { return __builtin_fmaf(a, c, b * s); }

// fw_qk_prep_fp8: the 8 results of a chunk as e4m3 bytes -- rounded to bf16 first (the value the bf16 form stores), then cast raw like
// fw_fp8_quant_rows(raw = 1), so the fused form returns the bits of the two-pass form.
__device__ __forceinline__ u32x2_t qk_pack_e4m3(const float* o) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = bf16_bits_to_f32(f32_to_bf16_bits(o[j]));
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    return u32x2_t{(uint32_t)lo, (uint32_t)hi};
}

// e4m3 output with a HEAD STRIDE (hs8 >= hd bytes per head; fw_qk_prep_fp8 head_stride8): chunk at element offset e (8 | e, never straddles
// a head) goes to head * hs8 + (e % hd); the chunk that ends a head also zero-fills the head's padding [hd, hs8) -- a head_dim-96
// operand laid out for the head_dim-128 fp8 attention kernel (zeros contribute nothing to q k^T).
__device__ __forceinline__ void qk_store_e4m3(uint8_t* row8, int e, int hd, int hs8, const float* o) {
    if (hs8 == hd) { *(u32x2_t*)(row8 + e) = qk_pack_e4m3(o); return; }          // no padding: contiguous heads
    // head index without an integer division (a long VALU sequence in a kernel that holds a whole row in registers): (e + 0.5) / hd
    // is at least 0.5 / hd away from an integer, far beyond the fp32 error of the product for e < 6144
    const int h = (int)(((float)e + 0.5f) * (1.0f / (float)hd)), w = e - h * hd;
    uint8_t* dst = row8 + h * hs8 + w;
    *(u32x2_t*)dst = qk_pack_e4m3(o);
    if (w + 8 == hd) {
        for (int z = hd; z < hs8; z += 8) *(u32x2_t*)(row8 + h * hs8 + z) = u32x2_t{0u, 0u};
    }
}

constexpr int QK_MAXC = 3;     // width <= 3 * 256 * 8 = 6144
constexpr int QK_MAXW = 6144;

__global__ __launch_bounds__(256) void qk_prep_kernel(uint16_t* __restrict__ x, int64_t ldx, int heads, int hd,
                                                      int norm_mode, const float* __restrict__ nw, const float* __restrict__ nb,
                                                      float eps, int rope_mode, const float* __restrict__ tab, int tab_rows, float oscale,
                                                      const float* __restrict__ ext_ss, int norm_width,
                                                      uint8_t* __restrict__ o8, int64_t ld8, int hs8) {
    __shared__ float rowbuf[QK_MAXW];
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int width = heads * hd;
    const int nch = width >> 3;
    uint16_t* xr = x + (int64_t)row * ldx;
    uint8_t* o8r = o8 ? o8 + (int64_t)row * ld8 : nullptr;          // e4m3 output (fw_qk_prep_fp8): x stays untouched
    float v[QK_MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < QK_MAXC; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
            const u32x4_t raw = *(const u32x4_t*)(xr + ch * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[i][2 * j] = __uint_as_float(raw[j] << 16);
                v[i][2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(v[i][j], v[i][j], ss);
        }
    }
    if (norm_mode == FW_NORM_RMS_FULL) {
        // ext_ss: the row's sum of squares over the FULL width (norm_width) when this call only holds a column slice of it
        // (head-sharded tensor parallelism: the partial sums are all-reduced between fw_row_sumsq and this call)
        const float tot = ext_ss ? ext_ss[row] : block_sum_256(ss, red);
        const float r = rsqrtf(tot / (float)(ext_ss ? norm_width : width) + eps);
#pragma unroll
        for (int i = 0; i < QK_MAXC; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = v[i][j] * r * nw[ch * 8 + j];
            }
        }
    } else if (norm_mode == FW_NORM_LN_HEAD) {
        // head_dim == 64: a head is 8 consecutive chunks = 8 consecutive lanes (chunk index = tid + 256 i)
#pragma unroll
        for (int i = 0; i < QK_MAXC; ++i) {
            const int ch = threadIdx.x + i * 256;
            float s1 = 0.f;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s1 += v[i][j];
            }
            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
            const float mean = s1 * (1.0f / 64.0f);
            float s2 = 0.f;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; s2 += d * d; }
            }
            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
            const float r = rsqrtf(s2 * (1.0f / 64.0f) + eps);
            if (ch < nch) {
                const int d0 = (ch * 8) & 63;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = __builtin_fmaf((v[i][j] - mean) * r, nw[d0 + j], nb[d0 + j]);
            }
        }
    }
    if (oscale != 1.0f) {      // rotations are linear: scaling the normalised row scales the result
#pragma unroll
        for (int i = 0; i < QK_MAXC; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] *= oscale;
    }
    if (rope_mode == FW_ROPE_NONE) {
#pragma unroll
        for (int i = 0; i < QK_MAXC; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nch) {
                if (o8r) { qk_store_e4m3(o8r, ch * 8, hd, hs8, v[i]); continue; }
                u32x4_t o = {pack_bf16x2(v[i][0], v[i][1]), pack_bf16x2(v[i][2], v[i][3]),
                             pack_bf16x2(v[i][4], v[i][5]), pack_bf16x2(v[i][6], v[i][7])};
                *(u32x4_t*)(xr + ch * 8) = o;
            }
        }
        return;
    }
    const float* trow = tab + (int64_t)(row % tab_rows) * hd;   // [hd/2][2]
    if (rope_mode == FW_ROPE_INTERLEAVED) {
        // partner is inside the same chunk
#pragma unroll
        for (int i = 0; i < QK_MAXC; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < nch) {
                const int e0 = (ch * 8) % hd;      // element offset inside the head
                float of[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float cs = trow[(e0 / 2 + j) * 2], sn = trow[(e0 / 2 + j) * 2 + 1];
                    const float a = v[i][2 * j], bq = v[i][2 * j + 1];
                    of[2 * j] = qk_rot_sub(a, cs, bq, sn);
                    of[2 * j + 1] = qk_rot_add(a, sn, bq, cs);
                }
                if (o8r) { qk_store_e4m3(o8r, ch * 8, hd, hs8, of); continue; }
                u32x4_t o4 = {pack_bf16x2(of[0], of[1]), pack_bf16x2(of[2], of[3]), pack_bf16x2(of[4], of[5]), pack_bf16x2(of[6], of[7])};
                *(u32x4_t*)(xr + ch * 8) = o4;
            }
        }
        return;
    }
    // FW_ROPE_HALF2D: head = [y half | x half], each half hd/2 wide, rotate-half pairs (w, w + hd/4)
#pragma unroll
    for (int i = 0; i < QK_MAXC; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rowbuf[ch * 8 + j] = v[i][j];
        }
    }
    __syncthreads();
    const int half = hd >> 1, quarter = hd >> 2;
#pragma unroll
    for (int i = 0; i < QK_MAXC; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
            const int base = ch * 8;
            const int e0 = base % hd;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + j;
                const int hsel = e / half, wi = e - hsel * half;
                const bool lo = wi < quarter;
                const int pidx = hsel * quarter + (lo ? wi : wi - quarter);
                const float cs = trow[pidx * 2], sn = trow[pidx * 2 + 1];
                const float partner = rowbuf[base + j + (lo ? quarter : -quarter)];
                o[j] = lo ? qk_rot_sub(v[i][j], cs, partner, sn) : qk_rot_add(v[i][j], cs, partner, sn);
            }
            if (o8r) { qk_store_e4m3(o8r, base, hd, hs8, o); continue; }
            u32x4_t o4 = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
            *(u32x4_t*)(xr + base) = o4;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Fast path of qk_prep: ONE WAVE per row, chunk (8 channels, 16 B) index = lane + 64*i, statistics and the rotate-half
// partner exchange by wave shuffles (head_dim 64: a head is 8 consecutive lanes, the partner chunk is lane ^ 2).
// ------------------------------------------------------------------------------------------------------------
// OUT8: fw_qk_prep_fp8 (e4m3 to a side buffer) -- its own instantiations, so that the bf16 forms keep the register count they had
// (the head-strided e4m3 store pushed <10, 1, 1> from 256 to 280 registers: one wave per SIMD on an HBM-bound pass)
// (OUT8: 0 = bf16 in place, 1 = e4m3 with contiguous heads, 2 = e4m3 with a head stride and zero padding)
template <int CPL, int NORM, int ROPE, int OUT8>
__global__ __launch_bounds__(256) void qk_prep_wave_kernel(uint16_t* __restrict__ x, int64_t ldx, int rows, int heads, int hd,
                                                           const float* __restrict__ nw, const float* __restrict__ nb, float eps,
                                                           const float* __restrict__ tab, int tab_rows, float oscale,
                                                           const float* __restrict__ ext_ss, int norm_width,
                                                           uint8_t* __restrict__ o8, int64_t ld8, int hs8) {
    const int lane = threadIdx.x & 63;
    const int width = heads * hd;
    const int nch = width >> 3;
    // Full-width RMSNorm (the DiT q / k: 5120 weights = 20 KiB for every 10 KiB row): the weight vector is staged ONCE per work-group
    // in LDS and the work-groups walk the rows (the launcher caps the grid), instead of re-reading it through the texture path per
    // row -- the LayerNorm kernel's finding (tools/probes/stream_bw.py).  Other modes: one row per wave, the loop runs once.
    constexpr bool STAGE = NORM == FW_NORM_RMS_FULL;
    __shared__ f32x4_t snw[STAGE ? CPL * 128 : 1];
    if (STAGE) {
        for (int i = threadIdx.x; i < nch * 2; i += 256) snw[i] = *(const f32x4_t*)(nw + i * 4);
        __syncthreads();
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    uint16_t* xr = x + (int64_t)row * ldx;
    float v[CPL][8];
    u32x4_t raw[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = lane + 64 * i;
        raw[i] = u32x4_t{0, 0, 0, 0};
        if (ch < nch) raw[i] = *(const u32x4_t*)(xr + ch * 8);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i][2 * j] = __uint_as_float(raw[i][j] << 16);
            v[i][2 * j + 1] = __uint_as_float(raw[i][j] & 0xffff0000u);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(v[i][j], v[i][j], ss);
    }
    if (NORM == FW_NORM_RMS_FULL) {
        const float r = ext_ss ? rsqrtf(ext_ss[row] / (float)norm_width + eps) : rsqrtf(wave_sum(ss) / (float)width + eps);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                const f32x4_t w0 = snw[STAGE ? ch * 2 : 0], w1 = snw[STAGE ? ch * 2 + 1 : 0];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][j] = v[i][j] * r * w0[j]; v[i][4 + j] = v[i][4 + j] * r * w1[j]; }
            }
        }
    } else if (NORM == FW_NORM_LN_HEAD) {
        // head_dim == 64: 8 consecutive chunks = 8 consecutive lanes
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int ch = lane + 64 * i;
            float s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += v[i][j];
            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
            const float mean = s1 * (1.0f / 64.0f);
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; s2 += d * d; }
            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
            const float r = rsqrtf(s2 * (1.0f / 64.0f) + eps);
            const int d0 = (ch * 8) & 63;
            const f32x4_t w0 = *(const f32x4_t*)(nw + d0), w1 = *(const f32x4_t*)(nw + d0 + 4);
            const f32x4_t b0 = *(const f32x4_t*)(nb + d0), b1 = *(const f32x4_t*)(nb + d0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[i][j] = __builtin_fmaf((v[i][j] - mean) * r, w0[j], b0[j]);
                v[i][4 + j] = __builtin_fmaf((v[i][4 + j] - mean) * r, w1[j], b1[j]);
            }
        }
    }
    if (oscale != 1.0f) {      // rotations are linear: scaling the normalised row scales the result
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] *= oscale;
    }
    // the row's rotary table ([hd/2][2] fp32 = 512 B at hd 128) is shared by every head of the row: one 8-byte load per lane puts it
    // into a per-wave LDS strip, and the 20 lookups per lane go to LDS instead of through the texture path
    __shared__ float strow[4][128];
    const float* trow = nullptr;
    if (ROPE != FW_ROPE_NONE) {
        const float* grow = tab + (int64_t)(row % tab_rows) * hd;   // [hd/2][2]
        float* mine = strow[threadIdx.x >> 6];
        if (lane * 2 < hd) *(u32x2_t*)(mine + lane * 2) = *(const u32x2_t*)(grow + lane * 2);
        trow = mine;
        // the strip is written and read by the lanes of ONE wave: a wavefront-scope release/acquire pair + wave barrier keeps the
        // compiler from moving the cross-lane reads above the store (lockstep alone is not a language guarantee)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = lane + 64 * i;
        float o[8];
        if (ROPE == FW_ROPE_INTERLEAVED) {
            const int e0 = (ch * 8) % hd;
            f32x4_t t0 = {1.f, 0.f, 1.f, 0.f}, t1 = {1.f, 0.f, 1.f, 0.f};
            if (ch < nch) { t0 = *(const f32x4_t*)(trow + e0); t1 = *(const f32x4_t*)(trow + e0 + 4); }
            o[0] = qk_rot_sub(v[i][0], t0[0], v[i][1], t0[1]); o[1] = qk_rot_add(v[i][0], t0[1], v[i][1], t0[0]);
            o[2] = qk_rot_sub(v[i][2], t0[2], v[i][3], t0[3]); o[3] = qk_rot_add(v[i][2], t0[3], v[i][3], t0[2]);
            o[4] = qk_rot_sub(v[i][4], t1[0], v[i][5], t1[1]); o[5] = qk_rot_add(v[i][4], t1[1], v[i][5], t1[0]);
            o[6] = qk_rot_sub(v[i][6], t1[2], v[i][7], t1[3]); o[7] = qk_rot_add(v[i][6], t1[3], v[i][7], t1[2]);
        } else if (ROPE == FW_ROPE_HALF2D) {
            // hd == 64: chunk k = ch & 7 of the head; y half = chunks 0..3, x half = 4..7; inside a half the first two
            // chunks are the "lo" 16 elements, partner chunk = k ^ 2 (lane ^ 2); table pair index = hsel*16 + (k&1)*8 + j
            const int k = ch & 7;
            const bool lo = (k & 2) == 0;
            const int pidx = (k >> 2) * 16 + (k & 1) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float partner = __shfl_xor(v[i][j], 2, 64);
                float cs = 1.f, sn = 0.f;
                if (ch < nch) { cs = trow[(pidx + j) * 2]; sn = trow[(pidx + j) * 2 + 1]; }
                o[j] = lo ? qk_rot_sub(v[i][j], cs, partner, sn) : qk_rot_add(v[i][j], cs, partner, sn);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[i][j];
        }
        if (ch < nch) {
            if (OUT8 == 1) {        // fw_qk_prep_fp8: e4m3 bytes to the side buffer, x untouched
                *(u32x2_t*)(o8 + (int64_t)row * ld8 + ch * 8) = qk_pack_e4m3(o);
            } else if (OUT8 == 2) {
                qk_store_e4m3(o8 + (int64_t)row * ld8, ch * 8, hd, hs8, o);
            } else {
                u32x4_t o4 = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                *(u32x4_t*)(xr + ch * 8) = o4;
            }
        }
    }
    if (ROPE != FW_ROPE_NONE) {     // this row's strip reads are done before the next row's strip store
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    }       // row loop
}

template <int CPL>
static bool launch_qk_wave(hipStream_t st, uint16_t* x, int64_t ldx, int rows, int heads, int hd, int norm, const float* nw,
                           const float* nb, float eps, int rope, const float* tab, int tab_rows, float oscale,
                           const float* ext_ss, int norm_width, uint8_t* o8, int64_t ld8, int hs8) {
    // RMS_FULL stages 20 KiB of weights per work-group and walks the rows: 4 work-groups per CU; the other modes: one row per wave
    const dim3 grid(norm == FW_NORM_RMS_FULL ? min((rows + 3) / 4, 256 * 4) : (rows + 3) / 4), block(256);
#define FW_QK_CASE(N, R) if (norm == N && rope == R) { \
        if (o8 && hs8 != hd) hipLaunchKernelGGL((qk_prep_wave_kernel<CPL, N, R, 2>), grid, block, 0, st, x, ldx, rows, heads, hd, nw, nb, eps, tab, tab_rows, oscale, ext_ss, norm_width, o8, ld8, hs8); \
        else if (o8) hipLaunchKernelGGL((qk_prep_wave_kernel<CPL, N, R, 1>), grid, block, 0, st, x, ldx, rows, heads, hd, nw, nb, eps, tab, tab_rows, oscale, ext_ss, norm_width, o8, ld8, hs8); \
        else hipLaunchKernelGGL((qk_prep_wave_kernel<CPL, N, R, 0>), grid, block, 0, st, x, ldx, rows, heads, hd, nw, nb, eps, tab, tab_rows, oscale, ext_ss, norm_width, o8, ld8, hs8); \
        return true; }
    FW_QK_CASE(FW_NORM_RMS_FULL, FW_ROPE_INTERLEAVED)
    FW_QK_CASE(FW_NORM_RMS_FULL, FW_ROPE_NONE)
    FW_QK_CASE(FW_NORM_NONE, FW_ROPE_INTERLEAVED)
    FW_QK_CASE(FW_NORM_LN_HEAD, FW_ROPE_HALF2D)
#undef FW_QK_CASE
    return false;
}

__global__ void sinusoid_kernel(const void* __restrict__ t, int t_dtype, float* __restrict__ out, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    double pos;
    if (t_dtype == FW_DT_BF16) pos = (double)bf16_bits_to_f32(*(const uint16_t*)t);
    else pos = (double)(*(const float*)t);
    const double f = pow(10000.0, -((double)i / (double)half));
    const double a = pos * f;
    out[i] = (float)cos(a);
    out[half + i] = (float)sin(a);
}

__device__ __forceinline__ float load_any(const void* p, int64_t i, int dtype) {
    return dtype == FW_DT_BF16 ? bf16_bits_to_f32(((const uint16_t*)p)[i]) : ((const float*)p)[i];
}

// one thread per (token l, channel c'): writes 4 bf16 (c'*4 .. c'*4+3); c' >= Cx+Cy writes the zero pad
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ x, int Cx, const void* __restrict__ y, int Cy, int dtype,
                                                       uint16_t* __restrict__ P, int64_t ldp, int F, int H2, int W2, int cols4) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Hh = H2 >> 1, Ww = W2 >> 1;
    const int64_t L = (int64_t)F * Hh * Ww;
    if (gid >= L * cols4) return;
    const int c = (int)(gid % cols4);
    const int64_t l = gid / cols4;
    const int w = (int)(l % Ww), h = (int)((l / Ww) % Hh), f = (int)(l / ((int64_t)Ww * Hh));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < Cx + Cy) {
        const void* src = c < Cx ? x : y;
        const int cc = c < Cx ? c : c - Cx;
        const int64_t base = (((int64_t)cc * F + f) * H2 + 2 * h) * W2 + 2 * w;
        v[0] = load_any(src, base, dtype); v[1] = load_any(src, base + 1, dtype);
        v[2] = load_any(src, base + W2, dtype); v[3] = load_any(src, base + W2 + 1, dtype);
    }
    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2_t*)(P + l * ldp + c * 4) = o;
}

// one thread per (token l, column): column = (y*2+z)*16 + c  (patch (1,2,2), 16 output channels)
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ Hd, int64_t ldh, void* __restrict__ out, int out_dtype,
                                                         int F, int Hh, int Ww) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t L = (int64_t)F * Hh * Ww;
    if (gid >= L * 64) return;
    const int col = (int)(gid & 63);
    const int64_t l = gid >> 6;
    const int w = (int)(l % Ww), h = (int)((l / Ww) % Hh), f = (int)(l / ((int64_t)Ww * Hh));
    const int c = col & 15, z = (col >> 4) & 1, yy = (col >> 5) & 1;
    const float v = Hd[l * ldh + col];
    const int64_t o = (((int64_t)c * F + f) * (2 * Hh) + 2 * h + yy) * (2 * Ww) + 2 * w + z;
    if (out_dtype == FW_DT_F32) ((float*)out)[o] = v;
    else ((uint16_t*)out)[o] = f32_to_bf16_bits(v);
}


// Wan2.2 control adapter front end (wan_video_camera_controller.py:24-44): PixelUnshuffle(8) followed by Conv2d(k=2, s=2) is a
// GEMM over 16x16 pixel patches.  in: [C][F][16h][16w]; out P[L][C*256] bf16 with column (c*64 + dy*8 + dx)*4 + ky*2 + kx
// (= Conv2d weight [N][C*64][2][2] flattened) = in[c][f][(2h+ky)*8 + dy][(2w+kx)*8 + dx].  One thread per (token, c, dy):
// two 16-element input rows in, 32 contiguous bf16 out.  Runs once per generation (the result is cached by the engine).
__global__ __launch_bounds__(256) void control_patchify_kernel(const void* __restrict__ in, int dtype, uint16_t* __restrict__ P,
                                                               int64_t ldp, int C, int F, int Hh, int Ww) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t L = (int64_t)F * Hh * Ww;
    if (gid >= L * C * 8) return;
    const int dy = (int)(gid & 7);
    const int c = (int)((gid >> 3) % C);
    const int64_t l = gid / (8 * (int64_t)C);
    const int w = (int)(l % Ww), h = (int)((l / Ww) % Hh), f = (int)(l / ((int64_t)Ww * Hh));
    const int Hp = 16 * Hh, Wp = 16 * Ww;
    uint16_t o[32];
#pragma unroll
    for (int ky = 0; ky < 2; ++ky) {
        const int64_t base = (((int64_t)c * F + f) * Hp + (2 * h + ky) * 8 + dy) * Wp + 16 * w;
#pragma unroll
        for (int kx = 0; kx < 2; ++kx)
#pragma unroll
            for (int dx = 0; dx < 8; ++dx) o[dx * 4 + ky * 2 + kx] = f32_to_bf16_bits(load_any(in, base + kx * 8 + dx, dtype));
    }
    uint16_t* dst = P + l * ldp + ((int64_t)c * 64 + dy * 8) * 4;
#pragma unroll
    for (int j = 0; j < 32; j += 2) *(uint32_t*)(dst + j) = (uint32_t)o[j] | ((uint32_t)o[j + 1] << 16);
}

// im2col for a 3x3, pad 1, stride 1 convolution on token-major activations (wan_video_camera_controller.py:64-76
// ResidualBlock): x [F][h][w][C] bf16 -> out [L][C*9], column c*9 + ky*3 + kx (= Conv2d weight [N][C][3][3] flattened)
// = x[f][hh+ky-1][ww+kx-1][c], zero outside the frame.  One thread per (token, channel pair); once per generation.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                        int64_t ldo, int C, int F, int Hh, int Ww) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t L = (int64_t)F * Hh * Ww;
    const int c2 = C >> 1;
    if (gid >= L * c2) return;
    const int c = (int)(gid % c2) * 2;
    const int64_t l = gid / c2;
    const int w = (int)(l % Ww), h = (int)((l / Ww) % Hh);
    uint16_t v[18];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hh = h + ky - 1, ww = w + kx - 1;
            uint32_t pair = 0;
            if (hh >= 0 && hh < Hh && ww >= 0 && ww < Ww) pair = *(const uint32_t*)(x + (l + (int64_t)(ky - 1) * Ww + (kx - 1)) * ldx + c);
            v[ky * 3 + kx] = (uint16_t)(pair & 0xffffu);
            v[9 + ky * 3 + kx] = (uint16_t)(pair >> 16);
        }
    uint16_t* dst = out + l * ldo + (int64_t)c * 9;      // c even -> 4-byte aligned
#pragma unroll
    for (int j = 0; j < 18; j += 2) *(uint32_t*)(dst + j) = (uint32_t)v[j] | ((uint32_t)v[j + 1] << 16);
}

__global__ __launch_bounds__(256) void assemble_tokens_kernel(const uint16_t* __restrict__ patch, int64_t ldp,
                                                              const float* __restrict__ special, float* __restrict__ tokens,
                                                              int S, int hw, int n_special, int C) {
    const int P = n_special + hw;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 4 channels
    const int c4 = C >> 2;
    if (gid >= (int64_t)S * P * c4) return;
    const int cc = (int)(gid % c4) * 4;
    const int64_t tp = gid / c4;
    const int pidx = (int)(tp % P);
    const int s = (int)(tp / P);
    f32x4_t v;
    if (pidx < n_special) {
        v = *(const f32x4_t*)(special + ((int64_t)(s == 0 ? 0 : 1) * n_special + pidx) * C + cc);
    } else {
        const u32x2_t raw = *(const u32x2_t*)(patch + ((int64_t)s * hw + (pidx - n_special)) * ldp + cc);
        v[0] = __uint_as_float(raw[0] << 16); v[1] = __uint_as_float(raw[0] & 0xffff0000u);
        v[2] = __uint_as_float(raw[1] << 16); v[3] = __uint_as_float(raw[1] & 0xffff0000u);
    }
    *(f32x4_t*)(tokens + tp * C + cc) = v;
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                                                            int rows, int C) {
    const int c4 = C >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)rows * c4) return;
    const int cc = (int)(gid % c4) * 4;
    const int64_t r = gid / c4;
    const f32x4_t v = *(const f32x4_t*)(x + r * ldx + cc);
    u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2_t*)(y + r * ldy + cc) = o;
}

}  // namespace

extern "C" int fw_layernorm_mod(const void* x, int64_t ldx, int x_dtype, uint16_t* y, int64_t ldy, int rows, int C,
                                const float* w, const float* b, const float* scale, const float* shift, float eps, void* stream) {
    if (rows <= 0) return 0;
    if (C <= 0 || (C % 4) || C > LN_MAXV * 256 * 4 || (ldx % 4) || (ldy % 4)) { fw_set_error("fw_layernorm_mod: C % 4 == 0, C <= 8192, ld % 4 == 0 required"); return FW_E_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype != FW_DT_F32 && x_dtype != FW_DT_BF16) { fw_set_error("fw_layernorm_mod: bad x_dtype"); return FW_E_BADARG; }
    {   // wave-per-row fast path: needs 16-B aligned rows and parameter vectors
        const bool xal = x_dtype == FW_DT_F32 ? ((((uintptr_t)x) & 15) == 0) : ((((uintptr_t)x) & 7) == 0);
        const bool pal = ((((uintptr_t)w) | ((uintptr_t)b) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0;
        if ((C % 256) == 0 && xal && pal && ((((uintptr_t)y) & 7) == 0)) {
            const bool ok = x_dtype == FW_DT_F32 ? launch_ln_wave<true>(st, x, ldx, y, ldy, rows, C, w, b, scale, shift, eps)
                                                 : launch_ln_wave<false>(st, x, ldx, y, ldy, rows, C, w, b, scale, shift, eps);
            if (ok) return (int)hipGetLastError();
        }
    }
    if (x_dtype == FW_DT_F32) hipLaunchKernelGGL(layernorm_mod_kernel<true>, dim3(rows), dim3(256), 0, st, x, ldx, y, ldy, C, w, b, scale, shift, eps);
    else if (x_dtype == FW_DT_BF16) hipLaunchKernelGGL(layernorm_mod_kernel<false>, dim3(rows), dim3(256), 0, st, x, ldx, y, ldy, C, w, b, scale, shift, eps);
    return (int)hipGetLastError();
}

// LayerNorm whose output keeps what the bf16 rounding drops: y = y_hi + y_lo, both bf16 (round 6; the per-site ablation of the bf16
// floor found ONE store worth a third of it -- the LayerNorm in front of the output head, whose rounding reaches noise_pred with no
// residual stream to average it: docs/parity.md).  The consumer runs its (tiny) GEMM on both parts.  fp32 input, C % 256 == 0.
extern "C" int fw_layernorm_mod_split(const float* x, int64_t ldx, uint16_t* y_hi, uint16_t* y_lo, int64_t ldy, int rows, int C,
                                      const float* w, const float* b, const float* scale, const float* shift, float eps, void* stream) {
    if (rows <= 0) return 0;
    const bool pal = ((((uintptr_t)w) | ((uintptr_t)b) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0;
    if (!x || !y_hi || !y_lo || C <= 0 || (C % 256) || (ldx % 4) || (ldy % 4) || (((uintptr_t)x) & 15) || ((((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 7) || !pal) {
        fw_set_error("fw_layernorm_mod_split: fp32 x, C % 256 == 0 (1024 / 1280 / 2048 / 5120), 16-B aligned rows and parameter vectors required"); return FW_E_BADARG; }
    if (!launch_ln_wave<true>((hipStream_t)stream, x, ldx, y_hi, ldy, rows, C, w, b, scale, shift, eps, y_lo)) {
        fw_set_error("fw_layernorm_mod_split: unsupported width"); return FW_E_UNSUPPORTED; }
    return (int)hipGetLastError();
}

static int qk_prep_impl(uint16_t* x, int64_t ldx, int rows, int heads, int head_dim, int norm_mode, const float* norm_w,
                        const float* norm_b, float eps, int rope_mode, const float* rope_tab, int tab_rows, float out_scale,
                        const float* ext_ss, int norm_width, void* stream, uint8_t* o8 = nullptr, int64_t ld8 = 0, int hs8 = 0) {
    if (rows <= 0) return 0;
    const int width = heads * head_dim;
    if (hs8 <= 0) hs8 = head_dim;
    if (o8 && ((ld8 % 8) || (((uintptr_t)o8) & 7) || hs8 < head_dim || (hs8 % 8))) {
        fw_set_error("fw_qk_prep_fp8: out8 8-byte aligned, ld8 % 8 == 0, head_stride8 >= head_dim and % 8 == 0 required"); return FW_E_BADARG; }
    if (width > QK_MAXW || (head_dim % 8) || (ldx % 8) || (((uintptr_t)x) & 15)) { fw_set_error("fw_qk_prep: width <= 6144, head_dim % 8 == 0, 16-B alignment required"); return FW_E_BADARG; }
    if (norm_mode == FW_NORM_LN_HEAD && (head_dim != 64 || !norm_w || !norm_b)) { fw_set_error("fw_qk_prep: LN_HEAD needs head_dim 64 and weight+bias"); return FW_E_BADARG; }
    if (norm_mode == FW_NORM_RMS_FULL && !norm_w) { fw_set_error("fw_qk_prep: RMS_FULL needs a weight"); return FW_E_BADARG; }
    if (ext_ss && (norm_mode != FW_NORM_RMS_FULL || norm_width < width)) { fw_set_error("fw_qk_prep_tp: external statistics need RMS_FULL and norm_width >= the slice width"); return FW_E_BADARG; }
    if (rope_mode != FW_ROPE_NONE && (!rope_tab || tab_rows <= 0)) { fw_set_error("fw_qk_prep: rope table missing"); return FW_E_BADARG; }
    if (rope_mode == FW_ROPE_HALF2D && (head_dim % 32)) { fw_set_error("fw_qk_prep: HALF2D needs head_dim % 32 == 0"); return FW_E_BADARG; }
    {   // wave-per-row fast path (table rows and norm vectors 16-B aligned; rotate-half form only for head_dim 64)
        const bool pal = ((((uintptr_t)norm_w) | ((uintptr_t)norm_b) | ((uintptr_t)rope_tab)) & 15) == 0;
        const bool shape_ok = (head_dim % 8) == 0 && (rope_mode != FW_ROPE_HALF2D || head_dim == 64) &&
                              (rope_mode != FW_ROPE_INTERLEAVED || (head_dim % 8) == 0);
        if (pal && shape_ok) {
            const int cpl = (width / 8 + 63) / 64;
            hipStream_t st = (hipStream_t)stream;
            bool ok = false;
            if (cpl <= 2) ok = launch_qk_wave<2>(st, x, ldx, rows, heads, head_dim, norm_mode, norm_w, norm_b, eps, rope_mode, rope_tab, tab_rows, out_scale, ext_ss, norm_width, o8, ld8, hs8);
            else if (cpl <= 3) ok = launch_qk_wave<3>(st, x, ldx, rows, heads, head_dim, norm_mode, norm_w, norm_b, eps, rope_mode, rope_tab, tab_rows, out_scale, ext_ss, norm_width, o8, ld8, hs8);
            else if (cpl <= 10) ok = launch_qk_wave<10>(st, x, ldx, rows, heads, head_dim, norm_mode, norm_w, norm_b, eps, rope_mode, rope_tab, tab_rows, out_scale, ext_ss, norm_width, o8, ld8, hs8);
            if (ok) return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(qk_prep_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ldx, heads, head_dim, norm_mode,
                       norm_w, norm_b, eps, rope_mode, rope_tab, tab_rows, out_scale, ext_ss, norm_width, o8, ld8, hs8);
    return (int)hipGetLastError();
}

extern "C" int fw_qk_prep(uint16_t* x, int64_t ldx, int rows, int heads, int head_dim, int norm_mode, const float* norm_w,
                          const float* norm_b, float eps, int rope_mode, const float* rope_tab, int tab_rows, float out_scale,
                          void* stream) {
    return qk_prep_impl(x, ldx, rows, heads, head_dim, norm_mode, norm_w, norm_b, eps, rope_mode, rope_tab, tab_rows, out_scale,
                        nullptr, 0, stream);
}

extern "C" int fw_qk_prep_tp(uint16_t* x, int64_t ldx, int rows, int heads, int head_dim, const float* norm_w, float eps,
                             int rope_mode, const float* rope_tab, int tab_rows, float out_scale,
                             const float* row_sumsq, int norm_width, void* stream) {
    if (!row_sumsq) { fw_set_error("fw_qk_prep_tp: row_sumsq missing"); return FW_E_BADARG; }
    return qk_prep_impl(x, ldx, rows, heads, head_dim, FW_NORM_RMS_FULL, norm_w, nullptr, eps, rope_mode, rope_tab, tab_rows,
                        out_scale, row_sumsq, norm_width, stream);
}

extern "C" int fw_qk_prep_fp8(const uint16_t* x, int64_t ldx, int rows, int heads, int head_dim, int norm_mode, const float* norm_w,
                              const float* norm_b, float eps, int rope_mode, const float* rope_tab, int tab_rows, float out_scale,
                              const float* row_sumsq, int norm_width, uint8_t* out8, int64_t ld8, int head_stride8, void* stream) {
    if (!out8) { fw_set_error("fw_qk_prep_fp8: out8 missing"); return FW_E_BADARG; }
    // x is only read when out8 is given (the kernels branch on it before every store)
    return qk_prep_impl(const_cast<uint16_t*>(x), ldx, rows, heads, head_dim, norm_mode, norm_w, norm_b, eps, rope_mode, rope_tab,
                        tab_rows, out_scale, row_sumsq, row_sumsq ? norm_width : 0, stream, out8, ld8, head_stride8);
}

// ---- per-forward modulation tables (fw_modulation_tables): one launch for all blocks of a kind --------------------------------------
__global__ __launch_bounds__(256) void modulation_tables_kernel(const float* __restrict__ mod, const float* __restrict__ t, int t_rows,
                                                                const float* __restrict__ ls2, float* __restrict__ table,
                                                                float* __restrict__ g1, float* __restrict__ g0, int rows, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* m = mod + (int64_t)b * rows * C;
    float* o = table + (int64_t)b * rows * C;
    float e3 = 0.f, e4 = 0.f, e5 = 0.f;
    for (int r = 0; r < rows; ++r) {
        const float v = m[(int64_t)r * C + c] + t[(int64_t)(r % t_rows) * C + c];
        o[(int64_t)r * C + c] = v;
        if (r == 3) e3 = v;
        if (r == 4) e4 = v;
        if (r == 5) e5 = v;
    }
    if (ls2) {      // the tensor ops this replaces: ls2 * (1.0 + e[4]) * e[5] and ls2 * e[3] * e[5], left to right, every product rounded
        const float l = ls2[(int64_t)b * C + c];
        g1[(int64_t)b * C + c] = (l * (1.0f + e4)) * e5;
        g0[(int64_t)b * C + c] = (l * e3) * e5;
    }
}

extern "C" int fw_modulation_tables(const float* mod, const float* t, int t_rows, const float* ls2, float* table, float* g1, float* g0,
                                    int nblk, int rows, int C, void* stream) {
    if (!mod || !t || !table || t_rows <= 0 || rows <= 0 || C <= 0 || (ls2 && (rows != 6 || !g1 || !g0))) {
        fw_set_error("fw_modulation_tables: mod / t / table required; ls2 needs rows == 6 and g1 / g0"); return FW_E_BADARG; }
    if (nblk <= 0) return 0;
    hipLaunchKernelGGL(modulation_tables_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       mod, t, t_rows, ls2, table, g1, g0, rows, C);
    return (int)hipGetLastError();
}

// ---- head-sharded tensor parallelism helpers (fantasy_world_amd/tensor_parallel.py) ---------------------------------------------
// per-row sum of squares of a bf16 slice, fp32: one wave per row
__global__ __launch_bounds__(256) void row_sumsq_kernel(const uint16_t* __restrict__ x, int64_t ldx, int rows, int width,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + (int64_t)row * ldx;
    float ss = 0.f;
    for (int c = lane * 8; c < width; c += 512) {
        const u32x4_t raw = *(const u32x4_t*)(xr + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = __uint_as_float(raw[j] << 16), b = __uint_as_float(raw[j] & 0xffff0000u);
            ss += a * a + b * b;
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) out[row] = ss;
}

extern "C" int fw_row_sumsq(const uint16_t* x, int64_t ldx, int rows, int width, float* out, void* stream) {
    if (rows <= 0) return 0;
    if ((width % 8) || (ldx % 8) || (((uintptr_t)x) & 15)) { fw_set_error("fw_row_sumsq: width % 8 == 0 and 16-B alignment required"); return FW_E_BADARG; }
    hipLaunchKernelGGL(row_sumsq_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, width, out);
    return (int)hipGetLastError();
}

// x[r][c] += (y[r][c] + bias[c]) * g1[c] + g0[c]: the epilogue of a row-parallel GEMM AFTER its partial sums were all-reduced
template <bool YF32>
__global__ __launch_bounds__(256) void residual_add_kernel(float* __restrict__ x, int64_t ldx, const void* __restrict__ yv, int64_t ldy,
                                                           int rows, int C, const float* __restrict__ bias,
                                                           const float* __restrict__ g1, const float* __restrict__ g0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (i >= (int64_t)rows * c4) return;
    const int r = (int)(i / c4), c = (int)(i % c4) * 4;
    float v[4];
    if (YF32) {
        const f32x4_t y = *(const f32x4_t*)((const float*)yv + (int64_t)r * ldy + c);
        v[0] = y[0]; v[1] = y[1]; v[2] = y[2]; v[3] = y[3];
    } else {
        const u32x2_t y = *(const u32x2_t*)((const uint16_t*)yv + (int64_t)r * ldy + c);
        v[0] = __uint_as_float(y[0] << 16); v[1] = __uint_as_float(y[0] & 0xffff0000u);
        v[2] = __uint_as_float(y[1] << 16); v[3] = __uint_as_float(y[1] & 0xffff0000u);
    }
    f32x4_t xo = *(f32x4_t*)(x + (int64_t)r * ldx + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = v[j] + (bias ? bias[c + j] : 0.f);
        t = fw_affine(t, g1 ? g1[c + j] : 1.f, g0 ? g0[c + j] : 0.f);
        xo[j] += t;
    }
    *(f32x4_t*)(x + (int64_t)r * ldx + c) = xo;
}

extern "C" int fw_residual_add(float* x, int64_t ldx, const void* y, int64_t ldy, int y_dtype, int rows, int C,
                               const float* bias, const float* g1, const float* g0, void* stream) {
    if (rows <= 0) return 0;
    if ((C % 4) || (ldx % 4) || (ldy % 4) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 7) || (y_dtype != FW_DT_BF16 && y_dtype != FW_DT_F32)) {
        fw_set_error("fw_residual_add: C % 4 == 0, aligned rows, bf16 / f32 y required"); return FW_E_BADARG; }
    const int64_t n = (int64_t)rows * (C / 4);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (y_dtype == FW_DT_F32) hipLaunchKernelGGL(residual_add_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C, bias, g1, g0);
    else hipLaunchKernelGGL(residual_add_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C, bias, g1, g0);
    return (int)hipGetLastError();
}

// ---- sampler step on the device -------------------------------------------------------------------------
// out = latents + (neg + s * (pos - neg)) * dsigma: the CFG combine (model_wan21.py:318-319) and the flow-match Euler update
// (flow_match.py:43-53) in ONE launch, with the roundings of the reference's five tensor ops: every intermediate is rounded to the
// tensors' dtype (bf16 tensors) / computed without contraction (fp32 tensors), so the result is BIT-identical to the PyTorch
// sequence.  params (optional, device): [s, dsigma] read by the kernel instead of the host values -- a captured HIP graph then
// replays with per-step values.
template <bool BF16>
__global__ __launch_bounds__(256) void cfg_euler_step_kernel(const void* __restrict__ posv, const void* __restrict__ negv,
                                                             const void* __restrict__ latv, void* __restrict__ outv, int64_t n,
                                                             float s, float ds, const float* __restrict__ params) {
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (params) { s = params[0]; ds = params[1]; }
    float p, q, x;
    if (BF16) {
        p = bf16_bits_to_f32(((const uint16_t*)posv)[i]); q = bf16_bits_to_f32(((const uint16_t*)negv)[i]);
        x = bf16_bits_to_f32(((const uint16_t*)latv)[i]);
        const float d = bf16_bits_to_f32(f32_to_bf16_bits(__fsub_rn(p, q)));
        const float e = bf16_bits_to_f32(f32_to_bf16_bits(__fmul_rn(d, s)));
        const float np = bf16_bits_to_f32(f32_to_bf16_bits(__fadd_rn(q, e)));
        const float m = bf16_bits_to_f32(f32_to_bf16_bits(__fmul_rn(np, ds)));
        ((uint16_t*)outv)[i] = f32_to_bf16_bits(__fadd_rn(x, m));
    } else {
        p = ((const float*)posv)[i]; q = ((const float*)negv)[i]; x = ((const float*)latv)[i];
        // five separately rounded fp32 operations, as five PyTorch kernels compute them (-ffast-math would contract them into FMAs)
        float d = p - q;
        asm volatile("" : "+v"(d));
        float e = d * s;
        asm volatile("" : "+v"(e));
        float np = q + e;
        asm volatile("" : "+v"(np));
        float m = np * ds;
        asm volatile("" : "+v"(m));
        ((float*)outv)[i] = x + m;
    }
}

extern "C" int fw_cfg_euler_step(const void* pos, const void* neg, const void* latents, void* out, int64_t n, int dtype,
                                 float cfg_scale, float dsigma, const float* dev_params, void* stream) {
    if (n <= 0) return 0;
    if (dtype != FW_DT_BF16 && dtype != FW_DT_F32) { fw_set_error("fw_cfg_euler_step: bf16 / f32 tensors"); return FW_E_BADARG; }
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == FW_DT_BF16) hipLaunchKernelGGL(cfg_euler_step_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, pos, neg, latents, out, n, cfg_scale, dsigma, dev_params);
    else hipLaunchKernelGGL(cfg_euler_step_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, pos, neg, latents, out, n, cfg_scale, dsigma, dev_params);
    return (int)hipGetLastError();
}

extern "C" int fw_sinusoid(const void* t, int t_dtype, float* out, int dim, void* stream) {
    if (dim <= 0 || (dim & 1) || (t_dtype != FW_DT_BF16 && t_dtype != FW_DT_F32)) { fw_set_error("fw_sinusoid: bad args"); return FW_E_BADARG; }
    hipLaunchKernelGGL(sinusoid_kernel, dim3((dim / 2 + 63) / 64), dim3(64), 0, (hipStream_t)stream, t, t_dtype, out, dim);
    return (int)hipGetLastError();
}

extern "C" int fw_patchify(const void* x, int Cx, const void* y, int Cy, int dtype, uint16_t* P, int64_t ldp,
                           int F, int H2, int W2, void* stream) {
    if ((H2 & 1) || (W2 & 1) || (ldp % 4) || ldp < 4 * (Cx + Cy) || (dtype != FW_DT_BF16 && dtype != FW_DT_F32)) { fw_set_error("fw_patchify: bad args"); return FW_E_BADARG; }
    const int cols4 = (int)(ldp / 4);
    const int64_t n = (int64_t)F * (H2 / 2) * (W2 / 2) * cols4;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, Cx, y, Cy, dtype, P, ldp, F, H2, W2, cols4);
    return (int)hipGetLastError();
}

extern "C" int fw_unpatchify(const float* Hd, int64_t ldh, void* out, int out_dtype, int F, int Hh, int Ww, void* stream) {
    if (out_dtype != FW_DT_BF16 && out_dtype != FW_DT_F32) { fw_set_error("fw_unpatchify: bad out_dtype"); return FW_E_BADARG; }
    const int64_t n = (int64_t)F * Hh * Ww * 64;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Hd, ldh, out, out_dtype, F, Hh, Ww);
    return (int)hipGetLastError();
}

extern "C" int fw_assemble_tokens(const uint16_t* patch, int64_t ldp, const float* special, float* tokens,
                                  int S, int hw, int n_special, int C, void* stream) {
    if ((C % 4) || (ldp % 4)) { fw_set_error("fw_assemble_tokens: C % 4 == 0 required"); return FW_E_BADARG; }
    const int64_t n = (int64_t)S * (n_special + hw) * (C / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, patch, ldp, special, tokens, S, hw, n_special, C);
    return (int)hipGetLastError();
}

extern "C" int fw_cast_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int C, void* stream) {
    if ((C % 4) || (ldx % 4) || (ldy % 4)) { fw_set_error("fw_cast_f32_bf16: C % 4 == 0 required"); return FW_E_BADARG; }
    const int64_t n = (int64_t)rows * (C / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C);
    return (int)hipGetLastError();
}

extern "C" int fw_control_patchify(const void* in, int dtype, uint16_t* P, int64_t ldp, int C, int F, int Hh, int Ww, void* stream) {
    if (C <= 0 || F <= 0 || Hh <= 0 || Ww <= 0 || (ldp % 2) || ldp < (int64_t)C * 256 || (((uintptr_t)P) & 3) ||
        (dtype != FW_DT_BF16 && dtype != FW_DT_F32)) { fw_set_error("fw_control_patchify: bad args"); return FW_E_BADARG; }
    const int64_t n = (int64_t)F * Hh * Ww * C * 8;
    hipLaunchKernelGGL(control_patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, dtype, P, ldp, C, F, Hh, Ww);
    return (int)hipGetLastError();
}

extern "C" int fw_im2col3x3(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int C, int F, int Hh, int Ww, void* stream) {
    if (C <= 0 || (C % 2) || (ldx % 2) || (ldo % 2) || ldo < (int64_t)C * 9 || (((uintptr_t)x) & 3) || (((uintptr_t)out) & 3)) {
        fw_set_error("fw_im2col3x3: C and leading dimensions must be even, bases 4-byte aligned"); return FW_E_BADARG; }
    const int64_t n = (int64_t)F * Hh * Ww * (C / 2);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, C, F, Hh, Ww);
    return (int)hipGetLastError();
}

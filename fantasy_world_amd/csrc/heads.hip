// Geometry-head kernels of libfw_mi355x.so: the data-movement and pointwise pieces around fw_gemm_bf16
// for the DPT heads, their temporal up-sampler and the camera head.  Every feature map is a channels-last matrix
// [frames*H*W][C] in bf16 (C a multiple of 8), so all of these are HBM-bound streaming kernels: 16-byte accesses along C,
// one work-item per 8 channels, no LDS.  They run once per generation (the last denoising step).
#include "fw_common.h"

namespace {

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}
__device__ __forceinline__ u32x4_t relu8(u32x4_t v) {
    // bf16 pairs: clear a half-word when its sign bit is set
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t w = v[i];
        if (w & 0x00008000u) w &= 0xffff0000u;
        if (w & 0x80000000u) w &= 0x0000ffffu;
        v[i] = w;
    }
    return v;
}

// Gather for a convolution as GEMM (Conv2d 3x3 stride 1/2, CausalConv3d (3,1,1) and 3x3x3: dpt_head.py:66-131,
// vae_modified.py:17-36).  out[(t-t0, yo, xo)][((dt*kh + dy)*kw + dx)*C + c] = x[t + dt - (kt-1)][yo*sh + dy - kh/2][xo*sw + dx - kw/2][c],
// zero outside the volume: causal in time, 'same' in space.  One work-item per (output row, tap, 8 channels).
__global__ __launch_bounds__(256) void im2col_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                     int64_t ldo, int C8, int T, int H, int W, int Ho, int Wo, int kt, int kh,
                                                     int kw, int sh, int sw, int ph, int pw, int up, int t0, int64_t total, int relu_in) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c8 = (int)(gid % C8);
    int64_t r = gid / C8;
    const int taps = kt * kh * kw;
    const int tap = (int)(r % taps);
    r /= taps;                                           // output row
    const int xo = (int)(r % Wo);
    const int yo = (int)((r / Wo) % Ho);
    const int tt = (int)(r / ((int64_t)Wo * Ho));
    const int dx = tap % kw, dy = (tap / kw) % kh, dt = tap / (kw * kh);
    const int ti = t0 + tt + dt - (kt - 1);
    const int yi = yo * sh + dy - ph;
    const int xi = xo * sw + dx - pw;
    // H, W are the sizes of the STORED map; with up = 2 the convolution sees its nearest-neighbour x2 up-sampling
    // (nn.Upsample(scale_factor=2, mode="nearest-exact"): source index = destination index / 2), which is never materialised
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (ti >= 0 && ti < T && yi >= 0 && yi < H * up && xi >= 0 && xi < W * up) {
        v = *(const u32x4_t*)(x + (((int64_t)ti * H + yi / up) * W + xi / up) * ldx + c8 * 8);
        if (relu_in) v = relu8(v);
    }
    *(u32x4_t*)(out + r * ldo + ((int64_t)tap * C8 + c8) * 8) = v;
}

// F.interpolate(mode="bilinear", align_corners=True) on channels-last maps (dpt_head.py:538-566): source coordinate
// = dst * (in-1)/(out-1), fp32 blend of the four neighbours, one bf16 rounding.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                              int64_t ldo, int C8, int h, int w, int H, int W, float sy, float sx,
                                                              int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c8 = (int)(gid % C8);
    const int64_t r = gid / C8;
    const int X = (int)(r % W), Y = (int)((r / W) % H);
    const int64_t n = r / ((int64_t)W * H);
    const float fy = Y * sy, fx = X * sx;
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const uint16_t* base = x + (n * h * w) * ldx + c8 * 8;
    float a[8], b[8], c[8], d[8], o[8];
    unpack8(*(const u32x4_t*)(base + ((int64_t)y0 * w + x0) * ldx), a);
    unpack8(*(const u32x4_t*)(base + ((int64_t)y0 * w + x1) * ldx), b);
    unpack8(*(const u32x4_t*)(base + ((int64_t)y1 * w + x0) * ldx), c);
    unpack8(*(const u32x4_t*)(base + ((int64_t)y1 * w + x1) * ldx), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float top = a[i] + (b[i] - a[i]) * wx, bot = c[i] + (d[i] - c[i]) * wx;
        o[i] = top + (bot - top) * wy;
    }
    *(u32x4_t*)(out + r * ldo + c8 * 8) = pack8(o);
}

// SiLU(x / max(|x|_2, 1e-12) * sqrt(c_true) * gamma) over the channel axis (RMS_norm + SiLU of ResidualBlock_Half,
// vae_modified.py:39-54, 201-203).  One wave per row; padded channels are zero on input and stay zero (gamma padded with 0).
__global__ __launch_bounds__(256) void chan_rmsnorm_silu_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                                int64_t ldo, int64_t rows, int C, float scale,
                                                                const float* __restrict__ gamma, int silu) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* src = x + row * ldx;
    float ss = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        float f[8];
        unpack8(*(const u32x4_t*)(src + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
    }
    ss = wave_sum(ss);
    const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);
    uint16_t* dst = out + row * ldo;
    for (int c = lane * 8; c < C; c += 512) {
        float f[8];
        unpack8(*(const u32x4_t*)(src + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float y = f[i] * inv * gamma[c + i];
            f[i] = silu ? fw_silu(y) : y;
        }
        *(u32x4_t*)(dst + c) = pack8(f);
    }
}

// Same, for narrow maps (C = 64 / 128 / 256, e.g. the 96 -> 128-channel full-resolution level of the VAE decoder): a row only
// fills C/8 lanes, so one wave takes 64 / (C/8) rows and reduces inside each lane group.
template <int LPR>
__global__ __launch_bounds__(256) void chan_rmsnorm_silu_narrow_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                       uint16_t* __restrict__ out, int64_t ldo, int64_t rows,
                                                                       float scale, const float* __restrict__ gamma, int silu) {
    constexpr int RPW = 64 / LPR;                        // rows per wave
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int c = (lane % LPR) * 8;
    const bool live = row < rows;
    float f[8];
    if (live) unpack8(*(const u32x4_t*)(x + row * ldx + c), f);
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = 0.f;
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (!live) return;
    const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float y = f[i] * inv * gamma[c + i];
        f[i] = silu ? fw_silu(y) : y;
    }
    *(u32x4_t*)(out + row * ldo + c) = pack8(f);
}

// ConvTranspose2d with kernel = stride = k after its GEMM: y[(n, yy, xx)][(dy*k + dx)*C + c] -> out[(n, yy*k + dy, xx*k + dx)][c]
__global__ __launch_bounds__(256) void depth_to_space_kernel(const uint16_t* __restrict__ y, int64_t ldy, uint16_t* __restrict__ out,
                                                             int64_t ldo, int C8, int h, int w, int k, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c8 = (int)(gid % C8);
    const int64_t r = gid / C8;                           // output row
    const int W = w * k, H = h * k;
    const int X = (int)(r % W), Y = (int)((r / W) % H);
    const int64_t n = r / ((int64_t)W * H);
    const int xx = X / k, dx = X % k, yy = Y / k, dy = Y % k;
    const u32x4_t v = *(const u32x4_t*)(y + ((n * h + yy) * w + xx) * ldy + ((int64_t)(dy * k + dx) * C8 + c8) * 8);
    *(u32x4_t*)(out + r * ldo + c8 * 8) = v;
}

// x[(n, p)][c] += table[p][c]: the UV positional embedding, identical for every frame (dpt_head.py:262-283)
__global__ __launch_bounds__(256) void add_table_kernel(uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ table,
                                                        int C8, int hw, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c8 = (int)(gid % C8);
    const int64_t r = gid / C8;
    const float* t = table + (r % hw) * (int64_t)(C8 * 8) + c8 * 8;
    uint16_t* p = x + r * ldx + c8 * 8;
    float f[8];
    unpack8(*(const u32x4_t*)p, f);
    const f32x4_t t0 = *(const f32x4_t*)t, t1 = *(const f32x4_t*)(t + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] += t0[i];
        f[4 + i] += t1[i];
    }
    *(u32x4_t*)p = pack8(f);
}

// temporal up-sampler: y[(i, p)][j*C + c] -> out[(2i + j, p)][c]  (vae_modified.py:121-124)
__global__ __launch_bounds__(256) void unfold_time2_kernel(const uint16_t* __restrict__ y, int64_t ldy, uint16_t* __restrict__ out,
                                                           int64_t ldo, int C8, int hw, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c8 = (int)(gid % C8);
    const int64_t r = gid / C8;                           // output row = (2i + j)*hw + p
    const int64_t p = r % hw, f = r / hw;
    const int64_t i = f >> 1, j = f & 1;
    *(u32x4_t*)(out + r * ldo + c8 * 8) = *(const u32x4_t*)(y + (i * hw + p) * ldy + (j * C8 + c8) * 8);
}

// out = relu?(a + b) on contiguous bf16 tensors (b may be null): the FeatureFusionBlock sum that the next
// ResidualConvUnit's in-place ReLU rewrites (dpt_head.py:517-522, 440)
__global__ __launch_bounds__(256) void add_act_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                      uint16_t* __restrict__ out, int64_t n8, int relu) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n8) return;
    float fa[8], fb[8];
    unpack8(*(const u32x4_t*)(a + gid * 8), fa);
    if (b) {
        unpack8(*(const u32x4_t*)(b + gid * 8), fb);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] += fb[i];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = fmaxf(fa[i], 0.f);
    }
    *(u32x4_t*)(out + gid * 8) = pack8(fa);
}

// camera head AdaLN (camera_head.py:124-128): out = gate * (LN(x) * (1 + scale) + shift) + x, per-row shift | scale | gate in
// mod[row][3C], no affine, fp32 throughout.  One wave per row.
__global__ __launch_bounds__(256) void adaln_rows_kernel(const float* __restrict__ x, const float* __restrict__ mod,
                                                         float* __restrict__ out, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = x + (int64_t)row * C;
    float s = 0.f, ss = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = src[c];
        s += v;
        ss += v * v;
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const float mean = s / C;
    const float rstd = rsqrtf(fmaxf(ss / C - mean * mean, 0.f) + eps);
    const float* m = mod + (int64_t)row * 3 * C;
    for (int c = lane; c < C; c += 64) {
        const float v = src[c];
        out[(int64_t)row * C + c] = m[2 * C + c] * ((v - mean) * rstd * (1.f + m[C + c]) + m[c]) + v;
    }
}

// activate_head / activate_pose (head_act.py:11-33, 61-125): y [rows][n] fp32.
//   mode 0 "exp":     pts = exp(y[:, :n-1]),                       conf = 1 + exp(y[:, n-1])
//   mode 1 "inv_log": pts = sign(v) * expm1(|v|),                  conf = 1 + exp(y[:, n-1])
//   mode 2 "pose":    pts[:, :n] = y with ReLU on columns >= 7 (translation, quaternion linear; field of view ReLU)
__global__ __launch_bounds__(256) void head_activation_kernel(const float* __restrict__ y, int64_t rows, int n, int mode,
                                                              float* __restrict__ pts, float* __restrict__ conf) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* src = y + r * n;
    if (mode == 2) {
        for (int c = 0; c < n; ++c) pts[r * n + c] = c >= 7 ? fmaxf(src[c], 0.f) : src[c];
        return;
    }
    for (int c = 0; c < n - 1; ++c) {
        const float v = src[c];
        pts[r * (n - 1) + c] = mode == 0 ? expf(v) : copysignf(expm1f(fabsf(v)), v);
    }
    conf[r] = 1.f + expf(src[n - 1]);
}

// nn.PixelUnshuffle(r) on a channels-last image stack (pose_adaptor_ac3d.py:26,91): in [F][H][W][C] (f32 or bf16) ->
// out [(f, y, x)][c*r*r + dy*r + dx] = in[f][y*r + dy][x*r + dx][c], bf16.  One work-item per output element pair.
template <typename T>
__global__ __launch_bounds__(256) void pixel_unshuffle_kernel(const T* __restrict__ in, uint16_t* __restrict__ out, int64_t ldo,
                                                              int H, int W, int C, int r, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int K = C * r * r;
    const int col = (int)(gid % (K / 2)) * 2;
    const int64_t row = gid / (K / 2);
    const int w = W / r, h = H / r;
    const int x = (int)(row % w), y = (int)((row / w) % h);
    const int64_t f = row / ((int64_t)w * h);
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cc = col + j;
        const int c = cc / (r * r), dy = (cc / r) % r, dx = cc % r;
        const T e = in[((f * H + (y * r + dy)) * W + (x * r + dx)) * C + c];
        if constexpr (sizeof(T) == 2) v[j] = bf16_bits_to_f32(e); else v[j] = e;
    }
    *(uint32_t*)(out + row * ldo + col) = pack_bf16x2(v[0], v[1]);
}

// nn.GroupNorm(G, C) on channels-last rows [frames*hw][C] (pose_adaptor_ac3d.py:30-40): statistics per (frame, group) over
// hw x C/G values, biased variance, then y = (x - mean) * rstd * w[c] + b[c] (+ ReLU).  One work-group per (frame, group):
// the slab (a few MB at most) is read twice, fp32 accumulation.
__global__ __launch_bounds__(1024) void group_norm_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                               int64_t ldo, int hw, int C, int G, const float* __restrict__ w,
                                                               const float* __restrict__ b, float eps, int relu) {
    __shared__ float red[2][16];
    const int f = blockIdx.x / G, g = blockIdx.x % G;
    const int cg = C / G, c8n = cg / 8;
    const uint16_t* src = x + (int64_t)f * hw * ldx + g * cg;
    const int64_t n8 = (int64_t)hw * c8n;
    float s = 0.f, ss = 0.f;
    for (int64_t i = threadIdx.x; i < n8; i += 1024) {
        const int64_t p = i / c8n;
        const int c8 = (int)(i % c8n);
        float v[8];
        unpack8(*(const u32x4_t*)(src + p * ldx + c8 * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += v[j]; ss += v[j] * v[j]; }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    s = 0.f; ss = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s += red[0][j]; ss += red[1][j]; }
    const float cnt = (float)hw * (float)cg;
    const float mean = s / cnt;
    const float rstd = rsqrtf(fmaxf(ss / cnt - mean * mean, 0.f) + eps);
    uint16_t* dst = out + (int64_t)f * hw * ldo + g * cg;
    for (int64_t i = threadIdx.x; i < n8; i += 1024) {
        const int64_t p = i / c8n;
        const int c8 = (int)(i % c8n);
        float v[8];
        unpack8(*(const u32x4_t*)(src + p * ldx + c8 * 8), v);
        const float* wp = w + g * cg + c8 * 8;
        const float* bp = b + g * cg + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (v[j] - mean) * rstd * wp[j] + bp[j];
            if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        *(u32x4_t*)(dst + p * ldo + c8 * 8) = pack8(v);
    }
}

// CameraPoseEncoder.compress_time (pose_adaptor_ac3d.py:61-76) on rows [frames*hw][C]: odd frame count keeps frame 0 and
// averages frames (2i+1, 2i+2); even averages (2i, 2i+1).  One work-item per 8 channels of an output row.
__global__ __launch_bounds__(256) void time_avg_pool_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                            int64_t ldo, int C8, int hw, int odd, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c8 = (int)(gid % C8);
    const int64_t r = gid / C8;
    const int64_t p = r % hw, fo = r / hw;
    u32x4_t res;
    if (odd && fo == 0) {
        res = *(const u32x4_t*)(x + p * ldx + c8 * 8);
    } else {
        const int64_t fa = odd ? 2 * fo - 1 : 2 * fo;
        float a[8], b[8];
        unpack8(*(const u32x4_t*)(x + (fa * hw + p) * ldx + c8 * 8), a);
        unpack8(*(const u32x4_t*)(x + ((fa + 1) * hw + p) * ldx + c8 * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (a[j] + b[j]) * 0.5f;
        res = pack8(a);
    }
    *(u32x4_t*)(out + r * ldo + c8 * 8) = res;
}

// out = act(x) on a contiguous bf16 tensor (the GELU between LayerNorm and Linear in CameraPoseEncoder.fc, pose_adaptor_ac3d.py:43-48)
__global__ __launch_bounds__(256) void activation_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int64_t n8, int act) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n8) return;
    float v[8];
    unpack8(*(const u32x4_t*)(x + gid * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fw_apply_act(v[j], act);
    *(u32x4_t*)(out + gid * 8) = pack8(v);
}

// Row softmax of an fp32 score matrix (the single 384-wide head of the VAE's AttentionBlock, wan_video_vae.py:262-268, is two
// GEMMs around this): out[r][c] = exp((s[r][c] - max_r) * scale) / sum_r for c < cols, 0 for cols <= c < cols_pad, bf16.
// One work-group per row: the row (<= a few ten thousand scores) is read twice, from L2 the second time.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds_, uint16_t* __restrict__ out,
                                                           int64_t ldo, int cols, int cols_pad, float scale) {
    __shared__ float red[4];
    const float* src = s + (int64_t)blockIdx.x * lds_;
    uint16_t* dst = out + (int64_t)blockIdx.x * ldo;
    float m = -3.0e38f;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, src[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) sum += __expf((src[c] - m) * scale);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = threadIdx.x * 2; c < cols_pad; c += 512) {
        const float a = c < cols ? __expf((src[c] - m) * scale) * inv : 0.f;
        const float b = c + 1 < cols ? __expf((src[c + 1] - m) * scale) * inv : 0.f;
        *(uint32_t*)(dst + c) = pack_bf16x2(a, b);
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int fw_im2col(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int C, int T, int H, int W, int kt, int kh,
                         int kw, int sh, int sw, int ph, int pw, int up, int t0, int nt, int relu_in, void* stream) {
    if (C <= 0 || (C % 8) || (ldx % 8) || (ldo % 8) || !aligned16(x) || !aligned16(out)) {
        fw_set_error("fw_im2col: C, ldx, ldo must be multiples of 8 and the bases 16-byte aligned"); return FW_E_BADARG; }
    if (kt < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || up < 1 || t0 < 0 || nt < 0 || t0 + nt > T ||
        H * up + 2 * ph < kh || W * up + 2 * pw < kw) {
        fw_set_error("fw_im2col: bad kernel / stride / padding / up-sampling / frame window"); return FW_E_BADARG; }
    const int Ho = (H * up + 2 * ph - kh) / sh + 1, Wo = (W * up + 2 * pw - kw) / sw + 1;
    const int64_t total = (int64_t)nt * Ho * Wo * kt * kh * kw * (C / 8);
    if (total <= 0) return 0;
    if ((total + 255) / 256 > 0x7fffffffLL) { fw_set_error("fw_im2col: grid too large, chunk the frames"); return FW_E_BADARG; }
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, C / 8, T, H, W,
                       Ho, Wo, kt, kh, kw, sh, sw, ph, pw, up, t0, total, relu_in);
    return (int)hipGetLastError();
}

extern "C" int fw_resize_bilinear(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int N, int h, int w, int H, int W,
                                  int C, void* stream) {
    if (C <= 0 || (C % 8) || (ldx % 8) || (ldo % 8) || !aligned16(x) || !aligned16(out) || h < 1 || w < 1 || H < 1 || W < 1) {
        fw_set_error("fw_resize_bilinear: C, ldx, ldo must be multiples of 8, bases 16-byte aligned, sizes positive"); return FW_E_BADARG; }
    const int64_t total = (int64_t)N * H * W * (C / 8);
    if (total <= 0) return 0;
    if ((total + 255) / 256 > 0x7fffffffLL) { fw_set_error("fw_resize_bilinear: grid too large, chunk the frames"); return FW_E_BADARG; }
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, C / 8, h, w,
                       H, W, sy, sx, total);
    return (int)hipGetLastError();
}

extern "C" int fw_chan_rmsnorm_silu(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int64_t rows, int C, int c_true,
                                    const float* gamma, int silu, void* stream) {
    if (C <= 0 || (C % 8) || (ldx % 8) || (ldo % 8) || !aligned16(x) || !aligned16(out) || c_true <= 0 || c_true > C || !gamma) {
        fw_set_error("fw_chan_rmsnorm_silu: bad arguments"); return FW_E_BADARG; }
    if (rows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const float sc = sqrtf((float)c_true);
    if (C == 64 || C == 128 || C == 256) {
        const int lpr = C / 8, rpb = 4 * (64 / lpr);     // rows per 256-thread block
        const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
        if (C == 64) hipLaunchKernelGGL(chan_rmsnorm_silu_narrow_kernel<8>, dim3(grid), dim3(256), 0, st, x, ldx, out, ldo, rows, sc, gamma, silu);
        else if (C == 128) hipLaunchKernelGGL(chan_rmsnorm_silu_narrow_kernel<16>, dim3(grid), dim3(256), 0, st, x, ldx, out, ldo, rows, sc, gamma, silu);
        else hipLaunchKernelGGL(chan_rmsnorm_silu_narrow_kernel<32>, dim3(grid), dim3(256), 0, st, x, ldx, out, ldo, rows, sc, gamma, silu);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(chan_rmsnorm_silu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ldx, out, ldo,
                       rows, C, sc, gamma, silu);
    return (int)hipGetLastError();
}

extern "C" int fw_depth_to_space(const uint16_t* y, int64_t ldy, uint16_t* out, int64_t ldo, int N, int h, int w, int k, int C,
                                 void* stream) {
    if (C <= 0 || (C % 8) || (ldy % 8) || (ldo % 8) || !aligned16(y) || !aligned16(out) || k < 1) {
        fw_set_error("fw_depth_to_space: bad arguments"); return FW_E_BADARG; }
    const int64_t total = (int64_t)N * h * k * w * k * (C / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(depth_to_space_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, ldy, out, ldo, C / 8, h, w, k,
                       total);
    return (int)hipGetLastError();
}

extern "C" int fw_add_table(uint16_t* x, int64_t ldx, const float* table, int64_t rows, int hw, int C, void* stream) {
    if (C <= 0 || (C % 8) || (ldx % 8) || !aligned16(x) || !aligned16(table) || hw <= 0 || (rows % hw)) {
        fw_set_error("fw_add_table: C, ldx % 8, 16-byte bases and rows % hw == 0 required"); return FW_E_BADARG; }
    const int64_t total = rows * (C / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(add_table_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, table, C / 8, hw, total);
    return (int)hipGetLastError();
}

extern "C" int fw_unfold_time2(const uint16_t* y, int64_t ldy, uint16_t* out, int64_t ldo, int n, int hw, int C, void* stream) {
    if (C <= 0 || (C % 8) || (ldy % 8) || (ldo % 8) || !aligned16(y) || !aligned16(out)) {
        fw_set_error("fw_unfold_time2: bad arguments"); return FW_E_BADARG; }
    const int64_t total = (int64_t)2 * n * hw * (C / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(unfold_time2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, ldy, out, ldo, C / 8, hw, total);
    return (int)hipGetLastError();
}

extern "C" int fw_add_act(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, int relu, void* stream) {
    if ((n % 8) || !aligned16(a) || !aligned16(out) || (b && !aligned16(b))) {
        fw_set_error("fw_add_act: element count must be a multiple of 8 and the bases 16-byte aligned"); return FW_E_BADARG; }
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_act_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 8, relu);
    return (int)hipGetLastError();
}

extern "C" int fw_adaln_rows(const float* x, const float* mod, float* out, int rows, int C, float eps, void* stream) {
    if (!x || !mod || !out || C <= 0) { fw_set_error("fw_adaln_rows: bad arguments"); return FW_E_BADARG; }
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(adaln_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, mod, out, rows, C, eps);
    return (int)hipGetLastError();
}

extern "C" int fw_head_activation(const float* y, int64_t rows, int n, int mode, float* pts, float* conf, void* stream) {
    if (!y || !pts || n < 2 || mode < 0 || mode > 2 || (mode != 2 && !conf)) { fw_set_error("fw_head_activation: bad arguments"); return FW_E_BADARG; }
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(head_activation_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, y, rows, n, mode, pts, conf);
    return (int)hipGetLastError();
}

extern "C" int fw_pixel_unshuffle(const void* in, int dtype, uint16_t* out, int64_t ldo, int F, int H, int W, int C, int r, void* stream) {
    if (!in || !out || r < 1 || (H % r) || (W % r) || ((C * r * r) % 2) || (ldo % 2) || (((uintptr_t)out) & 3) ||
        (dtype != FW_DT_BF16 && dtype != FW_DT_F32)) {
        fw_set_error("fw_pixel_unshuffle: H, W must be multiples of r, C*r*r and ldo even, dtype bf16 or f32"); return FW_E_BADARG; }
    const int64_t total = (int64_t)F * (H / r) * (W / r) * (C * r * r / 2);
    if (total <= 0) return 0;
    if (dtype == FW_DT_F32)
        hipLaunchKernelGGL(pixel_unshuffle_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)in, out,
                           ldo, H, W, C, r, total);
    else
        hipLaunchKernelGGL(pixel_unshuffle_kernel<uint16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)in, out, ldo, H, W, C, r, total);
    return (int)hipGetLastError();
}

extern "C" int fw_group_norm_rows(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int frames, int hw, int C, int groups,
                                  const float* w, const float* b, float eps, int relu, void* stream) {
    if (!w || !b || groups < 1 || C <= 0 || (C % groups) || ((C / groups) % 8) || (ldx % 8) || (ldo % 8) || !aligned16(x) || !aligned16(out)) {
        fw_set_error("fw_group_norm_rows: C/groups must be a multiple of 8, rows 16-byte aligned"); return FW_E_BADARG; }
    if (frames <= 0 || hw <= 0) return 0;
    hipLaunchKernelGGL(group_norm_rows_kernel, dim3((unsigned)(frames * groups)), dim3(1024), 0, (hipStream_t)stream, x, ldx, out, ldo,
                       hw, C, groups, w, b, eps, relu);
    return (int)hipGetLastError();
}

extern "C" int fw_time_avg_pool(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int frames, int hw, int C, void* stream) {
    if (C <= 0 || (C % 8) || (ldx % 8) || (ldo % 8) || !aligned16(x) || !aligned16(out) || frames < 1) {
        fw_set_error("fw_time_avg_pool: bad arguments"); return FW_E_BADARG; }
    const int odd = frames & 1;
    const int fout = odd ? 1 + (frames - 1) / 2 : frames / 2;
    const int64_t total = (int64_t)fout * hw * (C / 8);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(time_avg_pool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, C / 8, hw, odd, total);
    return (int)hipGetLastError();
}

extern "C" int fw_activation(const uint16_t* x, uint16_t* out, int64_t n, int act, void* stream) {
    if ((n % 8) || !aligned16(x) || !aligned16(out)) { fw_set_error("fw_activation: n % 8 == 0 and 16-byte bases required"); return FW_E_BADARG; }
    if (n <= 0) return 0;
    hipLaunchKernelGGL(activation_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, out, n / 8, act);
    return (int)hipGetLastError();
}

extern "C" int fw_softmax_rows(const float* s, int64_t lds, uint16_t* out, int64_t ldo, int rows, int cols, int cols_pad, float scale,
                               void* stream) {
    if (!s || !out || cols < 1 || cols_pad < cols || (cols_pad % 2) || (ldo % 2) || (((uintptr_t)out) & 3)) {
        fw_set_error("fw_softmax_rows: cols_pad >= cols, cols_pad and ldo even, 4-byte aligned output"); return FW_E_BADARG; }
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, lds, out, ldo, cols, cols_pad, scale);
    return (int)hipGetLastError();
}

"""Shared plumbing of the convolutional tails (geometry heads, VAE decoder): channels-last feature maps [frames*H*W, C] with C
padded to a multiple of 64, convolution weights flattened tap-major; a k x k (x k) convolution is ONE implicit-GEMM launch
(`fw_conv_gemm_bf16`: the tap gather is the LDS-DMA source address of the GEMM's A tile); `implicit_conv = False` on an instance
selects the older gather (`fw_im2col`) + GEMM pair, which computes the same bits and is kept as the A/B and test reference."""
import torch


def _cpad(c):
    return (c + 63) // 64 * 64


class ConvNetBase:
    """Needs `self.ops` and `self.max_col_bytes` (budget of one gathered matrix; frames are chunked to stay under it)."""

    def _lin(self, get, name, k_pad=None, n_pad=None):
        w = get(name + ".weight").float()
        w = w.reshape(w.shape[0], -1)
        b = get(name + ".bias").float()
        return self._pack(w, b, k_pad, n_pad)

    def _pack(self, w, b, k_pad=None, n_pad=None):
        N, K = w.shape
        k_pad, n_pad = k_pad or _cpad(K), n_pad or N
        wp = torch.zeros(n_pad, k_pad, dtype=torch.float32)
        wp[:N, :K] = w
        bp = None
        if b is not None:
            bp = torch.zeros(n_pad, dtype=torch.float32)
            bp[:N] = b
        return self.ops.pack_linear(wp, bp)

    def _conv(self, get, name, bias=True, pad_out=True):
        """Conv2d / Conv3d weight [N, C, *k] -> GEMM weight [N_pad, taps * C_pad], tap-major columns (the order `im2col` writes)."""
        w = get(name + ".weight").float()
        N, C = w.shape[:2]
        taps = 1
        for d in w.shape[2:]:
            taps *= d
        wt = w.reshape(N, C, taps).permute(0, 2, 1)                                  # [N, taps, C]
        cp = _cpad(C)
        wp = torch.zeros(N, taps, cp)
        wp[:, :, :C] = wt
        b = get(name + ".bias").float() if bias else None
        return self._pack(wp.reshape(N, taps * cp), b, taps * cp, _cpad(N) if pad_out else N)

    def _convT(self, get, name, k):
        """ConvTranspose2d(kernel = stride = k) weight [Cin, Cout, k, k] -> GEMM weight [(dy, dx, co), ci]."""
        w = get(name + ".weight").float()
        Ci, Co = w.shape[:2]
        cop = _cpad(Co)
        wp = torch.zeros(k, k, cop, _cpad(Ci))
        wp[:, :, :Co, :Ci] = w.permute(2, 3, 1, 0)
        bp = torch.zeros(k, k, cop)
        bp[:, :, :Co] = get(name + ".bias").float()
        return self.ops.pack_linear(wp.reshape(k * k * cop, _cpad(Ci)), bp.reshape(-1))

    def _vec(self, t, n_pad=None):
        t = t.float().reshape(-1)
        if n_pad is not None and n_pad != t.numel():
            t = torch.cat([t, torch.zeros(n_pad - t.numel(), device=t.device)])
        return self.ops.to_f32(t)

    def _conv_apply(self, x, T, H, W, lin, kt=1, kh=3, kw=3, sh=1, sw=1, act=None, res=None, relu_in=False, out_f32=False, up=1):
        """Convolution as gather + GEMM over frame chunks.  x [T*H*W, C] -> [T*Ho*Wo, N]; res (same rows) is added in the
        epilogue.  up = 2: the convolution runs on the nearest-neighbour x2 up-sampled map, which is never materialised."""
        ops = self.ops
        C = x.shape[1]
        Hu, Wu = H * up, W * up
        Ho, Wo = (Hu + 2 * (kh // 2) - kh) // sh + 1, (Wu + 2 * (kw // 2) - kw) // sw + 1
        rpf = Ho * Wo
        K = kt * kh * kw * C
        assert K == lin.K, (K, lin.K)
        if kt == 1 and kh == 1 and kw == 1 and up == 1:
            return ops.linear(x, lin, act=act, res=res, out_f32=out_f32)
        if C % 64 == 0 and not relu_in and getattr(self, "implicit_conv", True):
            # one implicit-GEMM launch per <= 255 frames: the gathered matrix is never written (fw_conv_gemm_bf16)
            if T <= 255 and (Ho - 1) * sh <= 4095 and (Wo - 1) * sw <= 4095:
                return ops.conv_gemm(x, T, H, W, lin, kt, kh, kw, sh=sh, sw=sw, up=up, act=act, res=res, out_f32=out_f32)
        step = max(1, int(self.max_col_bytes // (rpf * K * 2)))
        out = ops.empty(T * rpf, lin.N, dtype=torch.float32 if out_f32 else ops.act_dtype)
        for t0 in range(0, T, step):
            nt = min(step, T - t0)
            cols = ops.im2col(x, T, H, W, kt, kh, kw, sh, sw, t0, nt, relu_in, up=up)
            sl = slice(t0 * rpf, (t0 + nt) * rpf)
            ops.linear(cols, lin, act=act, res=None if res is None else res[sl], out_f32=out_f32, out=out[sl])
        return out

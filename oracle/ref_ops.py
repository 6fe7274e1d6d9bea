"""TEST INFRASTRUCTURE ONLY -- plain-PyTorch (CPU, fp32) implementation of the engine's op set.

Lets the host-side orchestration of fantasy_world_amd.engine (op order, weight packing/padding, modulation algebra,
rotary tables, sequence sharding) be checked on a machine without a GPU, against oracle/fw_oracle.py and the golden
fixtures.  It is never constructed by the product: fantasy_world_amd.install()/bench.py only ever build HipOps, which
raises when libfw_mi355x.so or the GPU is missing.  Each method restates the arithmetic of the matching C-ABI entry
point (include/fw_mi355x.h) and is what the `-m gpu` op tests compare the HIP kernels with.

`emulate_bf16=True` rounds activations to bf16 at the points where the HIP path stores bf16, which gives the
tolerance tests a realistic yardstick of the rounding the GPU path is allowed.
"""
import math

import torch
import torch.nn.functional as F


class _Lin:
    __slots__ = ("w", "b", "N", "K", "fp8")

    def __init__(self, w, b, fp8=False):
        self.w, self.b, self.fp8 = w, b, fp8
        self.N, self.K = w.shape


class TorchRefOps:
    name = "torch-ref"

    def __init__(self, emulate_bf16=False, device="cpu", exact=False):
        """exact: matrix products (linears, attention scores / PV) accumulate in fp64 and are rounded to fp32 once, so a result does
        not depend on how BLAS blocks M / N / K -- rows computed by a rank of a sharded run then carry the bits of the single-process
        run (needed where e4m3 rounding would amplify last-bit differences: the fp8 shard tests)."""
        self.exact = exact
        self.emulate_bf16 = emulate_bf16
        self.device = torch.device(device)
        self.act_dtype = torch.float32

    # ---- plumbing ---------------------------------------------------------------------------------------------
    # Per-site control of the bf16 emulation (WHICH bf16 stores make the 3.0e-3 floor?).  Every rounding point
    # names its site ("ln:C5120", "linear_out:N15360:K5120", "qk:hd128", "attn_o:hd128:long", "input", "cast_act", ...):
    #   fp32_sites = ("ln:", ...)   -> those sites stay fp32, everything else is rounded      (leave-one-out)
    #   only_sites = ("ln:", ...)   -> ONLY those sites are rounded, everything else stays fp32 (one-in)
    # prefixes; both empty = round everywhere (the yardstick of docs/parity.md).
    fp32_sites = ()
    only_sites = None
    emulate_p = False
    site_suffix = ":blk"        # appended to the LayerNorm site name; the ablation sets ":head" around the engine's epilogue

    def _r(self, t, site="other"):
        if not self.emulate_bf16:
            return t
        if self.only_sites is not None:
            if not site.startswith(tuple(self.only_sites)):
                return t
        elif self.fp32_sites and site.startswith(tuple(self.fp32_sites)):
            return t
        return t.to(torch.bfloat16).to(torch.float32)

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or torch.float32, device=self.device)

    def to_f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def to_act(self, t):
        return self._r(t.detach().to(device=self.device, dtype=torch.float32), "input").contiguous()

    def pack_linear(self, w, b, fp8=False):
        assert w.shape[1] % 64 == 0
        if fp8:
            return self.pack_linear_fp8(w, b)
        return _Lin(self._r(self.to_f32(w), "weight"), None if b is None else self.to_f32(b))

    def pack_linear_f32(self, w, b):
        return _Lin(self.to_f32(w), None if b is None else self.to_f32(b))

    # ---- ops --------------------------------------------------------------------------------------------------
    @staticmethod
    def _act(y, act):
        if act in (None, "none"):
            return y
        if act == "relu":
            return F.relu(y)
        if act == "gelu_tanh":
            return F.gelu(y, approximate="tanh")
        if act == "gelu_erf":
            return F.gelu(y)
        if act == "silu":
            return F.silu(y)
        raise ValueError(act)

    def linear(self, x, lin, act=None, g1=None, g0=None, res=None, out_f32=False, out=None):
        if lin.fp8:      # fp8_linear (layers.py:115-151) by its definition, then the same epilogue
            q, scale = x if isinstance(x, tuple) else self.quantize_fp8_rows(x)      # (rows already quantised, their scale)
            assert q.shape[1] == lin.K
            # e4m3 x e4m3 products summed in fp64: exact, so the result does not depend on how M / N / K are blocked or sharded
            y = ((q.double() @ lin.w.double().t()).float()) * scale[:, None]
        else:
            assert x.shape[1] == lin.K
            y = (x.double() @ lin.w.double().t()).float() if self.exact else x.to(torch.float32) @ lin.w.t()
        if lin.b is not None:
            y = y + lin.b
        y = self._act(y, act)
        if g1 is not None:
            y = y * g1
        if g0 is not None:
            y = y + g0
        if res is not None:
            y = y + res
        if not out_f32:
            y = self._r(y, f"linear_out:N{lin.N}:K{lin.K}")
        if out is not None:
            out.copy_(y)
            return out
        return y

    def linear_f32(self, x, lin, silu_in=False, act=None):
        x = x.reshape(-1).to(torch.float32)
        if silu_in:
            x = F.silu(x)
        y = (lin.w.double() @ x.double()).float() if self.exact else lin.w @ x
        if lin.b is not None:
            y = y + lin.b
        return self._act(y, act)

    def layernorm(self, x, w=None, b=None, scale=None, shift=None, eps=1e-6, out=None):
        y = F.layer_norm(x.to(torch.float32), (x.shape[-1],), w, b, eps)
        if scale is not None:
            y = y * (1.0 + scale)
        if shift is not None:
            y = y + shift
        return self._r(y, f"ln:C{x.shape[-1]}{self.site_suffix}")

    def layernorm_split(self, x, w=None, b=None, scale=None, shift=None, eps=1e-6):
        """fw_layernorm_mod_split: (hi, lo) with hi = bf16(y), lo = bf16(y - hi); without bf16 emulation hi = y, lo = 0."""
        y = F.layer_norm(x.to(torch.float32), (x.shape[-1],), w, b, eps)
        if scale is not None:
            y = y * (1.0 + scale)
        if shift is not None:
            y = y + shift
        site = f"ln:C{x.shape[-1]}{self.site_suffix}"
        hi = self._r(y, site)
        return hi, self._r(y - hi, site)

    def qk_prep(self, x, heads, hd, norm=None, norm_w=None, norm_b=None, eps=1e-6, rope=None, table=None, out_scale=1.0,
                ext_sumsq=None, norm_width=None, out8=None, head_stride8=None):
        if out8 is not None:          # fw_qk_prep_fp8 by its definition: cast_fp8(qk_prep(copy of x)), x untouched
            tmp = self.qk_prep(x.clone(), heads, hd, norm, norm_w, norm_b, eps, rope, table, out_scale, ext_sumsq, norm_width)
            hs = int(head_stride8 or hd)
            if hs == hd:
                return self.cast_fp8(tmp, out=out8)
            out8.zero_()              # every head zero-padded to head_stride8 bytes
            out8.view(x.shape[0], heads, hs)[:, :, :hd] = self.cast_fp8(tmp).view(x.shape[0], heads, hd)
            return out8
        rows = x.shape[0]
        v = x.to(torch.float32)
        if norm == "rms_full" and ext_sumsq is not None:         # head slice of a wider row: statistic supplied (fw_qk_prep_tp)
            v = v * torch.rsqrt(ext_sumsq.view(rows, 1) / float(norm_width) + eps) * norm_w
        elif norm == "rms_full":
            v = v * torch.rsqrt(v.pow(2).mean(dim=-1, keepdim=True) + eps) * norm_w
        elif norm == "ln_head":
            v = F.layer_norm(v.view(rows, heads, hd), (hd,), norm_w, norm_b, eps).reshape(rows, heads * hd)
        if rope not in (None, "none"):
            tab = table.to(x.device)[torch.arange(rows, device=x.device) % table.shape[0]]          # [rows, hd/2, 2]
            cs, sn = tab[..., 0].unsqueeze(1), tab[..., 1].unsqueeze(1)  # [rows, 1, hd/2]
            vh = v.view(rows, heads, hd)
            if rope == "interleaved":
                a, bq = vh[..., 0::2], vh[..., 1::2]
                o = torch.stack([a * cs - bq * sn, a * sn + bq * cs], dim=-1).reshape(rows, heads, hd)
            elif rope == "half2d":
                q4 = hd // 4
                vv = vh.view(rows, heads, 2, 2, q4)                      # [.., half(y/x), lo/hi, q4]
                c2 = cs.view(rows, 1, 2, q4)
                s2 = sn.view(rows, 1, 2, q4)
                lo, hi_ = vv[:, :, :, 0], vv[:, :, :, 1]
                o = torch.stack([lo * c2 - hi_ * s2, hi_ * c2 + lo * s2], dim=3).reshape(rows, heads, hd)
            else:
                raise ValueError(rope)
            v = o.reshape(rows, heads * hd)
        x.copy_(self._r(v * out_scale, f"qk:hd{hd}"))
        return x

    def row_sumsq(self, x, out=None):
        r = x.to(torch.float32).pow(2).sum(dim=-1)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def residual_add(self, x, y, bias=None, g1=None, g0=None):
        t = y.to(torch.float32)
        if bias is not None:
            t = t + bias
        if g1 is not None:
            t = t * g1
        if g0 is not None:
            t = t + g0
        x.add_(t)
        return x

    def cfg_euler_step(self, pos, neg, latents, cfg_scale, dsigma, out=None, dev_params=None):
        """The reference's tensor ops, literally (model_wan21.py:318-321, flow_match.py:52)."""
        if dev_params is not None:
            cfg_scale, dsigma = float(dev_params[0]), float(dev_params[1])
        noise_pred = neg + cfg_scale * (pos - neg)
        r = latents + noise_pred * torch.tensor(dsigma, dtype=torch.float32, device=latents.device)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def prepare_v(self, v, heads, hd, batch=1):
        return (v, v.shape[0] // batch)

    @staticmethod
    def q_scale(hd):
        return 1.4426950408889634 / math.sqrt(hd)

    def attention(self, q, k, v, heads, hd, batch=1, out=None, accumulate=False, v_prepared=None, q_prescaled=False):
        if v_prepared is not None:
            v = v_prepared[0]
        Lq, Lk = q.shape[0] // batch, k.shape[0] // batch
        qh = q.reshape(batch, Lq, heads, hd).transpose(1, 2).to(torch.float32)
        kh = k.reshape(batch, Lk, heads, hd).transpose(1, 2).to(torch.float32)
        vh = v.reshape(batch, Lk, heads, hd).transpose(1, 2).to(torch.float32)
        if self.exact:
            s = (qh.double() @ kh.double().transpose(-1, -2)) * (0.6931471805599453 if q_prescaled else 1.0 / math.sqrt(hd))
            o = (torch.softmax(s, dim=-1) @ vh.double()).float()
        elif self.emulate_bf16 and self.emulate_p:
            # the HIP kernels feed PV with P = 2^(s - m) ROUNDED to bf16; since round 6 the row sum adds the SAME rounded values (a ones
            # MFMA over the PV operand, attention.hip -- round 5 added the unrounded ones on the vector pipe): off in the default
            # yardstick (HIP = 1.003 x floor without it), switched on by the per-site ablation to price it
            s = (qh @ kh.transpose(-1, -2)) * (0.6931471805599453 if q_prescaled else 1.0 / math.sqrt(hd))
            pu = torch.exp(s - s.amax(dim=-1, keepdim=True))
            pr = self._r(pu, f"attn_p:hd{hd}")
            o = (pr @ vh) / pr.sum(dim=-1, keepdim=True)
        else:
            s = (qh @ kh.transpose(-1, -2)) * (0.6931471805599453 if q_prescaled else 1.0 / math.sqrt(hd))
            o = torch.softmax(s, dim=-1) @ vh
        o = o.transpose(1, 2).reshape(batch * Lq, heads * hd)
        if accumulate:
            o = o + out
        o = self._r(o, f"attn_o:hd{hd}:{'long' if Lk >= 1024 else 'short'}")
        if out is not None:
            out.copy_(o)
            return out
        return o

    def sinusoid(self, t, dim):
        half = dim // 2
        pos = t.reshape(-1)[:1].to(torch.float64)
        freq = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64, device=pos.device) / half)
        a = pos * freq
        return torch.cat([torch.cos(a), torch.sin(a)]).to(torch.float32)

    def patchify(self, x, y, kpad):
        xy = x if y is None else torch.cat([x, y.to(x.dtype)], dim=1)
        xy = xy.to(torch.float32)
        _, C, F_, H2, W2 = xy.shape
        h, w = H2 // 2, W2 // 2
        p = xy[0].view(C, F_, h, 2, w, 2).permute(1, 2, 4, 0, 3, 5).reshape(F_ * h * w, C * 4)
        out = torch.zeros(F_ * h * w, kpad, device=xy.device)
        out[:, : C * 4] = p
        return self._r(out, "input")

    def unpatchify(self, hd_out, F_, Hh, Ww, out_dtype):
        t = hd_out.view(F_, Hh, Ww, 1, 2, 2, 16)            # (f h w) (x y z c)
        t = t.permute(6, 0, 3, 1, 4, 2, 5).reshape(16, F_, 2 * Hh, 2 * Ww)
        return t.unsqueeze(0).to(out_dtype)

    def assemble_tokens(self, patch, special, S, hw):
        n_special, C = special.shape[1], special.shape[2]
        pt = patch.to(torch.float32).view(S, hw, C)
        sp = torch.stack([special[0 if s == 0 else 1] for s in range(S)], dim=0)
        return torch.cat([sp, pt], dim=1).reshape(S * (n_special + hw), C).contiguous()

    def control_patchify(self, ctl):
        _, C, F_, Hp, Wp = ctl.shape
        h, w = Hp // 16, Wp // 16
        u = F.pixel_unshuffle(ctl[0].to(torch.float32).permute(1, 0, 2, 3), 8)       # [F, C*64, 2h, 2w]
        p = u.reshape(F_, C * 64, h, 2, w, 2).permute(0, 2, 4, 1, 3, 5).reshape(F_ * h * w, C * 256)
        return self._r(p)

    def im2col3x3(self, x, F_, h, w):
        C = x.shape[1]
        img = x.to(torch.float32).view(F_, h, w, C).permute(0, 3, 1, 2)              # [F, C, h, w]
        cols = F.unfold(img, kernel_size=3, padding=1)                               # [F, C*9, h*w], row c*9 + ky*3 + kx
        return self._r(cols.transpose(1, 2).reshape(F_ * h * w, C * 9))

    def cast_act(self, x):
        return self._r(x.clone(), "cast_act")

    # ---- geometry heads: channels-last activations [T*H*W, C] ------------------------------------
    def im2col(self, x, T, H, W, kt, kh, kw, sh=1, sw=1, t0=0, nt=None, relu_in=False, ph=None, pw=None, up=1):
        """Gather for a convolution as GEMM: x [T*H*W, C] -> [nt*Ho*Wo, kt*kh*kw*C] for output frames t0..t0+nt-1, column
        ((dt*kh + dy)*kw + dx)*C + c = x[t + dt - (kt-1)][y*sh + dy - ph][x*sw + dx - pw][c], zero outside (causal in time:
        vae_modified.py:17-36; spatial padding ph / pw, default k//2 = 'same')."""
        C = x.shape[1]
        nt = T - t0 if nt is None else nt
        ph, pw = kh // 2 if ph is None else ph, kw // 2 if pw is None else pw
        v = x.to(torch.float32).view(T, H, W, C)
        if up != 1:      # gather from the nearest-neighbour up-sampled map (nn.Upsample(scale 2, 'nearest-exact'): src = dst // 2)
            v = v.repeat_interleave(up, dim=1).repeat_interleave(up, dim=2)
            H, W = H * up, W * up
        if relu_in:
            v = F.relu(v)
        v = F.pad(v, (0, 0, pw, pw, ph, ph, kt - 1, 0))
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        cols = []
        for dt in range(kt):
            for dy in range(kh):
                for dx in range(kw):
                    cols.append(v[t0 + dt:t0 + dt + nt, dy:dy + (Ho - 1) * sh + 1:sh, dx:dx + (Wo - 1) * sw + 1:sw])
        return self._r(torch.cat(cols, dim=-1).reshape(nt * Ho * Wo, kt * kh * kw * C))


    def conv_gemm(self, x, T, H, W, lin, kt, kh, kw, sh=1, sw=1, t0=0, nt=None, ph=None, pw=None, up=1, act=None, res=None,
                  out_f32=False, out=None):
        """The op the HIP side runs as one implicit-GEMM kernel (fw_conv_gemm_bf16): by definition gather + linear."""
        cols = self.im2col(x, T, H, W, kt, kh, kw, sh, sw, t0, nt, False, ph, pw, up)
        return self.linear(cols, lin, act=act, res=res, out_f32=out_f32, out=out)

    def resize_bilinear(self, x, N, h, w, H, W):
        """F.interpolate(mode='bilinear', align_corners=True) on [N*h*w, C] -> [N*H*W, C] (dpt_head.py:538-566)."""
        C = x.shape[1]
        v = x.to(torch.float32).view(N, h, w, C).permute(0, 3, 1, 2)
        v = F.interpolate(v, size=(H, W), mode="bilinear", align_corners=True)
        return self._r(v.permute(0, 2, 3, 1).reshape(N * H * W, C))

    def chan_rmsnorm_silu(self, x, gamma, c_true, silu=True):
        """SiLU(F.normalize(x, dim=channel) * sqrt(C) * gamma) (vae_modified.py:39-54, :201-203); padded channels are zero.
        silu=False: the bare RMS_norm of the VAE's AttentionBlock (wan_video_vae.py:246,256)."""
        v = x.to(torch.float32)
        y = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12) * (c_true ** 0.5) * gamma
        return self._r(F.silu(y) if silu else y)

    def softmax_rows(self, s, scale, cols_pad):
        """softmax(s * scale) over the columns of s fp32 [rows, cols] -> [rows, cols_pad] (zero beyond cols)."""
        p = torch.softmax(s.to(torch.float32) * scale, dim=-1)
        out = torch.zeros(s.shape[0], cols_pad, dtype=torch.float32, device=s.device)
        out[:, :s.shape[1]] = p
        return self._r(out)

    def depth_to_space(self, y, N, h, w, k, C):
        """[N*h*w, k*k*C] with column (dy*k + dx)*C + c -> [N*(h*k)*(w*k), C] (ConvTranspose2d with kernel = stride)."""
        v = y.view(N, h, w, k, k, C).permute(0, 1, 3, 2, 4, 5)
        return v.reshape(N * h * k * w * k, C).contiguous()

    def add_table(self, x, table):
        """x [N*hw, C] += table [hw, C] (fp32), in place (positional embedding, dpt_head.py:262-283)."""
        hw = table.shape[0]
        x.copy_(self._r((x.to(torch.float32).view(-1, hw, x.shape[1]) + table).view(x.shape)))
        return x

    def unfold_time2(self, y, n, hw, C):
        """[n*hw, 2C] -> [2n*hw, C]: columns [0:C] are frame 2i, [C:2C] frame 2i+1 (vae_modified.py:121-124)."""
        return y.view(n, hw, 2, C).permute(0, 2, 1, 3).reshape(2 * n * hw, C).contiguous()

    def add_act(self, a, b=None, relu=False):
        y = a.to(torch.float32) if b is None else a.to(torch.float32) + b.to(torch.float32)
        if relu:
            y = F.relu(y)
        return self._r(y)

    def adaln_rows(self, x, mod):
        """gate * (LN(x) * (1 + scale) + shift) + x with per-row shift|scale|gate = mod [rows, 3C] (camera_head.py:124-128)."""
        C = x.shape[1]
        shift, scale, gate = mod[:, :C], mod[:, C:2 * C], mod[:, 2 * C:]
        return gate * (F.layer_norm(x, (C,), None, None, 1e-6) * (1 + scale) + shift) + x

    def head_activation(self, y, mode):
        """activate_head / activate_pose (head_act.py): y fp32 [rows, n].  'exp' | 'inv_log': (pts [rows, n-1], 1+exp(conf));
        'pose': ReLU on columns 7.. (camera_head.py:38)."""
        if mode == "pose":
            return torch.cat([y[:, :7], F.relu(y[:, 7:])], dim=-1)
        xyz, conf = y[:, :-1], y[:, -1]
        pts = torch.exp(xyz) if mode == "exp" else torch.sign(xyz) * torch.expm1(xyz.abs())
        return pts, 1 + conf.exp()

    # ---- camera pose encoder -------------------------------------------------------------------------
    def pixel_unshuffle_rows(self, x, r):
        """x [F, H, W, C] -> [F*(H/r)*(W/r), C*r*r], column c*r*r + dy*r + dx (nn.PixelUnshuffle channel order)."""
        Fr, H, W, C = x.shape
        v = x.to(torch.float32).view(Fr, H // r, r, W // r, r, C).permute(0, 1, 3, 5, 2, 4)
        return self._r(v.reshape(Fr * (H // r) * (W // r), C * r * r))

    def group_norm_rows(self, x, frames, groups, w, b, eps=1e-5, relu=False):
        """nn.GroupNorm(groups, C) on channels-last rows: statistics per (frame, group) over (C/groups) x pixels."""
        rows, C = x.shape
        v = x.to(torch.float32).view(frames, rows // frames, groups, C // groups)
        mean = v.mean(dim=(1, 3), keepdim=True)
        var = v.var(dim=(1, 3), unbiased=False, keepdim=True)
        y = ((v - mean) * torch.rsqrt(var + eps)).view(rows, C) * w + b
        return self._r(F.relu(y) if relu else y)

    def time_avg_pool(self, x, frames, hw):
        """CameraPoseEncoder.compress_time (pose_adaptor_ac3d.py:61-76) on [frames*hw, C] rows -> (rows, new frame count)."""
        v = x.to(torch.float32).view(frames, hw, -1)
        if frames % 2 == 1:
            rest = v[1:]
            out = torch.cat([v[:1], (rest[0::2] + rest[1::2]) * 0.5], dim=0) if frames > 1 else v
        else:
            out = (v[0:frames - 1:2] + v[1:frames:2]) * 0.5
        return self._r(out.reshape(-1, v.shape[-1])), out.shape[0]

    def activation(self, x, act):
        return self._r(self._act(x.to(torch.float32), act))

    # ---- fp8 linear ----------------------------------------------------------------------------------------
    def pack_linear_fp8(self, w, b, bias_through_fp8=True):
        wq = self.to_f32(w).to(torch.bfloat16).to(torch.float8_e4m3fn)               # raw cast (layers.py:137)
        return _Lin(wq, None if b is None else self.fp8_bias(b, bias_through_fp8), fp8=True)

    def fp8_bias(self, b, bias_through_fp8=True):
        bb = self.to_f32(b).to(torch.bfloat16)
        if bias_through_fp8:          # AutoWrappedLinear.forward: cast_to(bias, computation_dtype) before fp8_linear (layers.py:158-159)
            bb = bb.to(torch.float8_e4m3fn).to(torch.bfloat16)
        return bb.to(torch.float32)

    def quantize_fp8_rows(self, x, amax=None):
        """AutoWrappedLinear.fp8_linear lines 126-136 on bf16-representable x [M, K]: (e4m3 tensor, fp32 scale [M]).
        amax: the row maximum supplied (x is a K-slice of a wider row; fw_fp8_quant_rows_amax)."""
        xb = x.to(torch.bfloat16)
        x_max = torch.max(torch.abs(xb), dim=-1, keepdim=True).values if amax is None else amax.to(torch.bfloat16).reshape(-1, 1)
        scale = torch.clamp(x_max / 448.0, min=1.0).float()
        return (xb / (scale + 1e-8)).to(torch.float8_e4m3fn), scale.reshape(-1)

    # ---- fp8 attention (include/fw_mi355x.h: fw_attention_fp8; PARITY UNPINNED -- the reference defines no fp8 attention).  A CPU
    # statement of the semantics the header states, so that the host orchestration (what is cast when, what the exchanges carry, which
    # heads a rank attends to) can be exercised over gloo without a GPU.  e4m3 tensors travel as uint8 views, like on the HIP side.
    FP8_Q_EXP = 3
    fp8_linear_exp = True         # the probabilities of attention_fp8: the piecewise-linear byte form (kernel default) or e4m3(exp2)

    def q_scale_fp8(self, hd):
        return self.q_scale(hd) * float(2 ** self.FP8_Q_EXP)

    def cast_fp8(self, x, out=None):
        q = x.to(torch.bfloat16).to(torch.float8_e4m3fn).view(torch.uint8)
        if out is not None:
            out.copy_(q)
            return out
        return q

    def prepare_v_fp8(self, v, heads, hd, batch=1, hd_out=None):
        v8 = v if v.dtype == torch.uint8 else self.cast_fp8(v)
        ho = int(hd_out or hd)
        if ho != hd:                  # rows hd .. hd_out-1 of every head: zeros
            pad = torch.zeros(v8.shape[0], heads, ho, dtype=torch.uint8, device=v8.device)
            pad[:, :, :hd] = v8.reshape(v8.shape[0], heads, hd)
            v8 = pad.reshape(v8.shape[0], heads * ho)
        return v8, v.shape[0] // batch

    def attention_fp8(self, q8, k8, vt8, heads, hd, Lk, batch=1, out=None):
        dec = lambda t: t.contiguous().view(torch.float8_e4m3fn).to(torch.float32)
        Lq = q8.shape[0] // batch
        qh = dec(q8).reshape(batch, Lq, heads, hd).transpose(1, 2) * float(2.0 ** -self.FP8_Q_EXP)
        kh = dec(k8).reshape(batch, Lk, heads, hd).transpose(1, 2)
        vh = dec(vt8).reshape(batch, Lk, heads, hd).transpose(1, 2)
        s = qh.double() @ kh.double().transpose(-1, -2)                 # log2-domain scores; e4m3 products: exact in fp64
        m = s.amax(dim=-1, keepdim=True)
        if self.fp8_linear_exp:      # round 6 default of the kernel: the e4m3 BYTE of P is round(8 (s - m + 7) + 56) -- 2^f ~ 1 + f inside a
            # binade, v_cvt_pk_u8_f32's round-to-nearest-even and saturation at 0 (csrc/attention_fp8.hip, tools/probes/cvt_pk_u8_probe.hip)
            # the shift is a WHOLE number of binades (the byte -> value map is exponential only from binade to binade): the largest
            # byte of a row lands in (104, 112]
            mq = 8.0 * torch.ceil((8.0 * m.float() - 112.0) / 8.0)
            bits = torch.round(8.0 * s.float() - mq).clamp(0, 255).to(torch.uint8)
            pr = bits.view(torch.float8_e4m3fn).to(torch.float64)
        else:                        # FW_ATTN_VAR=12: P = e4m3(2^(s - m + 7))
            pr = torch.exp2((s - m + 7.0).float()).to(torch.float8_e4m3fn).to(torch.float64)
        o = ((pr @ vh.double()) / pr.sum(dim=-1, keepdim=True)).float()  # numerator and normaliser from the SAME rounded P
        o = self._r(o.transpose(1, 2).reshape(batch * Lq, heads * hd))
        if out is not None:
            out.copy_(o)
            return out
        return o

    def row_absmax(self, x, out=None):
        r = x.to(torch.bfloat16).abs().amax(dim=-1).float()
        if out is not None:
            out.copy_(r)
            return out
        return r

    def modulation_tables(self, mod, t, ls2=None):
        """fw_modulation_tables: table[b][r] = mod[b][r] + t[r % t_rows]; with ls2 the fc2 epilogue's scale / offset per block."""
        nblk, rows, C = mod.shape
        t2 = t.reshape(-1, C)
        table = mod + t2[torch.arange(rows) % t2.shape[0]].unsqueeze(0)
        if ls2 is None:
            return table
        return table, ls2 * (1.0 + table[:, 4]) * table[:, 5], ls2 * table[:, 3] * table[:, 5]

    def linear_fp8(self, x, lin, out_f32=False):
        """torch._scaled_mm(xq, wq^T, scale_a, 1, bias, out_dtype) by its definition: (xq @ wq^T) * scale_a + bias."""
        return self.linear(x, lin, out_f32=out_f32)

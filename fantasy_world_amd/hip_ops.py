"""ctypes binding of libfw_mi355x.so (include/fw_mi355x.h) + the op set the engine is written against.

PyTorch is used here only as the owner of device memory and of the HIP stream; every arithmetic op of the
denoising forward goes through the C ABI.  There is NO fallback: if the shared library is missing, or no gfx950
device is visible, construction raises.
"""
import ctypes
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FW_LIB_PATH: A/B knob for kernel experiments (two builds of the library measured on one box); never a fallback
LIB_PATH = os.environ.get("FW_LIB_PATH") or os.path.join(_HERE, "libfw_mi355x.so")

FW_DT_NONE, FW_DT_BF16, FW_DT_F32 = 0, 1, 2
ACT = {None: 0, "none": 0, "relu": 1, "gelu_tanh": 2, "gelu_erf": 3, "silu": 4}
NORM = {None: 0, "none": 0, "rms_full": 1, "ln_head": 2}
ROPE = {None: 0, "none": 0, "interleaved": 1, "half2d": 2}

# every symbol declared in include/fw_mi355x.h (tests check the library exports all of them)
SYMBOLS = [
    "fw_abi_version", "fw_last_error", "fw_gemm_bf16", "fw_attention_bf16", "fw_attention_workspace_bytes", "fw_v_transpose", "fw_layernorm_mod",
    "fw_qk_prep", "fw_gemv_f32", "fw_sinusoid", "fw_patchify", "fw_unpatchify", "fw_assemble_tokens",
    "fw_cast_f32_bf16", "fw_set_option", "fw_debug_attention_timestamps", "fw_debug_gemm_timestamps", "fw_debug_gemm_pp_timestamps", "fw_control_patchify", "fw_im2col3x3",
    "fw_im2col", "fw_conv_gemm_bf16", "fw_v_transpose_fp8", "fw_attention_fp8", "fw_resize_bilinear", "fw_chan_rmsnorm_silu", "fw_depth_to_space", "fw_add_table", "fw_unfold_time2",
    "fw_add_act", "fw_adaln_rows", "fw_head_activation",
    "fw_pixel_unshuffle", "fw_group_norm_rows", "fw_time_avg_pool", "fw_activation", "fw_softmax_rows",
    "fw_fp8_quant_rows", "fw_gemm_fp8",
    "fw_row_sumsq", "fw_qk_prep_tp", "fw_residual_add", "fw_cfg_euler_step",
    "fw_qk_prep_fp8", "fw_v_transpose_e4m3", "fw_row_absmax", "fw_fp8_quant_rows_amax", "fw_modulation_tables",
    "fw_layernorm_mod_split",
]

_lib = None


def load_library(path: str = LIB_PATH, cache: bool = True):
    """dlopen the C-ABI library and declare argument types.  Raises if it is absent (no fallback).
    cache=False: a SECOND build of the library beside the process-wide one (tools/lib_ab.py: A/B of compiler flags in one process)."""
    global _lib
    if _lib is not None and cache:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The HIP path has no CPU fallback.")
    lib = ctypes.CDLL(path)
    c = ctypes
    vp, i64, i32, f32 = c.c_void_p, c.c_int64, c.c_int, c.c_float
    lib.fw_abi_version.restype = i32
    lib.fw_abi_version.argtypes = []
    lib.fw_last_error.restype = c.c_char_p
    lib.fw_last_error.argtypes = []
    sig = {
        "fw_gemm_bf16": [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp],
        "fw_attention_bf16": [vp, i64, i64, vp, i64, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, i32, f32, i32, vp, i64, vp],
        "fw_v_transpose": [vp, i64, i64, vp, i64, i32, i32, i32, i32, vp],
        "fw_layernorm_mod": [vp, i64, i32, vp, i64, i32, i32, vp, vp, vp, vp, f32, vp],
        "fw_qk_prep": [vp, i64, i32, i32, i32, i32, vp, vp, f32, i32, vp, i32, f32, vp],
        "fw_gemv_f32": [vp, vp, i64, vp, vp, i32, i32, i32, i32, vp],
        "fw_sinusoid": [vp, i32, vp, i32, vp],
        "fw_patchify": [vp, i32, vp, i32, i32, vp, i64, i32, i32, i32, vp],
        "fw_unpatchify": [vp, i64, vp, i32, i32, i32, i32, vp],
        "fw_assemble_tokens": [vp, i64, vp, vp, i32, i32, i32, i32, vp],
        "fw_cast_f32_bf16": [vp, i64, vp, i64, i32, i32, vp],
        "fw_set_option": [i32, i32],
        "fw_debug_attention_timestamps": [vp, i32],
        "fw_debug_gemm_timestamps": [vp, i32],
        "fw_debug_gemm_pp_timestamps": [vp, i32],
        "fw_control_patchify": [vp, i32, vp, i64, i32, i32, i32, i32, vp],
        "fw_im2col3x3": [vp, i64, vp, i64, i32, i32, i32, i32, vp],
        "fw_im2col": [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
        "fw_conv_gemm_bf16": [vp, i64] + [i32] * 14 + [vp, i64, vp, i64, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp],
        "fw_v_transpose_fp8": [vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, vp],
        "fw_attention_fp8": [vp, i64, i64, vp, i64, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, vp],
        "fw_softmax_rows": [vp, i64, vp, i64, i32, i32, i32, f32, vp],
        "fw_fp8_quant_rows": [vp, i64, vp, i64, vp, i32, i32, i32, vp],
        "fw_gemm_fp8": [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp],
        "fw_pixel_unshuffle": [vp, i32, vp, i64, i32, i32, i32, i32, i32, vp],
        "fw_group_norm_rows": [vp, i64, vp, i64, i32, i32, i32, i32, vp, vp, f32, i32, vp],
        "fw_time_avg_pool": [vp, i64, vp, i64, i32, i32, i32, vp],
        "fw_activation": [vp, vp, i64, i32, vp],
        "fw_resize_bilinear": [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp],
        "fw_chan_rmsnorm_silu": [vp, i64, vp, i64, i64, i32, i32, vp, i32, vp],
        "fw_depth_to_space": [vp, i64, vp, i64, i32, i32, i32, i32, i32, vp],
        "fw_add_table": [vp, i64, vp, i64, i32, i32, vp],
        "fw_unfold_time2": [vp, i64, vp, i64, i32, i32, i32, vp],
        "fw_add_act": [vp, vp, vp, i64, i32, vp],
        "fw_adaln_rows": [vp, vp, vp, i32, i32, f32, vp],
        "fw_head_activation": [vp, i64, i32, i32, vp, vp, vp],
        "fw_row_sumsq": [vp, i64, i32, i32, vp, vp],
        "fw_qk_prep_tp": [vp, i64, i32, i32, i32, vp, f32, i32, vp, i32, f32, vp, i32, vp],
        "fw_residual_add": [vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, vp],
        "fw_cfg_euler_step": [vp, vp, vp, vp, i64, i32, f32, f32, vp, vp],
        "fw_qk_prep_fp8": [vp, i64, i32, i32, i32, i32, vp, vp, f32, i32, vp, i32, f32, vp, i32, vp, i64, i32, vp],
        "fw_v_transpose_e4m3": [vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, vp],
        "fw_row_absmax": [vp, i64, i32, i32, vp, vp],
        "fw_fp8_quant_rows_amax": [vp, i64, vp, i64, vp, vp, i32, i32, vp],
        "fw_modulation_tables": [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp],
        "fw_layernorm_mod_split": [vp, i64, vp, vp, i64, i32, i32, vp, vp, vp, vp, f32, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = i32
        fn.argtypes = args
    lib.fw_attention_workspace_bytes.restype = i64
    lib.fw_attention_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    if lib.fw_abi_version() != 12:
        raise RuntimeError("libfw_mi355x.so ABI version mismatch")
    if cache:
        _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = _lib.fw_last_error().decode() if rc < 0 else f"hipError {rc}"
        raise RuntimeError(f"{what} failed: {msg}")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _dt(t):
    if t.dtype == torch.bfloat16:
        return FW_DT_BF16
    if t.dtype == torch.float32:
        return FW_DT_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


class Linear:
    """Pre-packed nn.Linear: bf16 weight [N, K] (K % 64 == 0, zero padded), fp32 bias [N].
    fp8 = True: the weight is e4m3 bytes (raw cast) and the bias holds bf16-rounded values -- AutoWrappedLinear.fp8_linear's
    operands (diffsynth_wan22/vram_management/layers.py:134-138); HipOps.linear then quantises the activation rows."""
    __slots__ = ("w", "b", "N", "K", "fp8")

    def __init__(self, w, b, fp8=False):
        self.w, self.b, self.fp8 = w, b, fp8
        self.N, self.K = w.shape


class LinearF32:
    __slots__ = ("w", "b", "N", "K")

    def __init__(self, w, b):
        self.w, self.b = w, b
        self.N, self.K = w.shape


class HipOps:
    """The engine's op set on one MI355X.  Activations are bf16; residual streams and statistics are fp32."""

    name = "hip"
    act_dtype = torch.bfloat16

    def __init__(self, device="cuda"):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs a visible gfx950 device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipOps only runs on a HIP device")
        self._timing = None
        self.split_kv = True           # split-KV for tail q-blocks that would cost an extra round (A/B: set False)

    # ---- live kernel timing (bench.py roofline): HIP events on the launch stream around selected launches -------------
    def start_kernel_timing(self, tags):
        """tags: {name: predicate(info)}; info = dict(kind="attention", hd, Lq, Lk, heads, batch) or
        dict(kind="linear", M, N, K, res, out_f32, act).  Every launch matching a predicate is bracketed by two HIP events
        recorded on the stream the kernel is launched on."""
        self._timing = {name: dict(pred=pred, events=[]) for name, pred in tags.items()}

    def stop_kernel_timing(self):
        """-> {name: (average launch duration in ms, number of launches timed)}; call after a device synchronize."""
        t, self._timing = self._timing, None
        out = {}
        for name, rec in (t or {}).items():
            ms = [a.elapsed_time(b) for a, b in rec["events"]]
            out[name] = (sum(ms) / len(ms), len(ms)) if ms else (0.0, 0)
        return out

    def _time_begin(self, info):
        if self._timing is None:
            return None
        hits = [rec for rec in self._timing.values() if rec["pred"](info)]
        if not hits:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(self.device))
        return hits, ev

    def _time_end(self, tok):
        if tok is not None:
            hits, ev = tok
            ev[1].record(torch.cuda.current_stream(self.device))
            for rec in hits:
                rec["events"].append(ev)

    # ---- memory plumbing ------------------------------------------------------------------------------------
    OPTS = {"gemm_tile": 0, "gemm_kernel": 1, "gemm_var": 2, "attn_var": 3}

    def set_option(self, name, value):
        """A/B knob (include/fw_mi355x.h FW_OPT_*); never changes results."""
        _check(self.lib.fw_set_option(self.OPTS[name], int(value)), "fw_set_option")

    def _stream(self):
        # a kernel launched on another device's stream is undefined behaviour: one process drives ONE GPU through this object
        if self.device.index is not None and torch.cuda.current_device() != self.device.index:
            raise RuntimeError(f"HipOps bound to {self.device} but the current HIP device is {torch.cuda.current_device()}: "
                               "call torch.cuda.set_device(...) (one process per GPU)")
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check_dev(self, *tensors):
        """Raw data_ptr() values cross the C ABI: a host tensor or one on another GPU must fail HERE, as a Python error."""
        for t in tensors:
            if t is not None and (not t.is_cuda or (self.device.index is not None and t.device.index != self.device.index)):
                raise RuntimeError(f"tensor on {t.device} passed to HipOps bound to {self.device}")

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.act_dtype, device=self.device)

    def to_f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def to_act(self, t):
        return t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()

    def pack_linear(self, w, b, fp8=False):
        """w fp32/bf16 [N, K] with K % 64 == 0 (engine pads), b [N] or None.  fp8: pack for the fp8 linear instead."""
        if fp8:
            return self.pack_linear_fp8(w, b)
        assert w.shape[1] % 64 == 0, w.shape
        wb = w.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
        bb = None if b is None else b.detach().to(device=self.device, dtype=torch.float32).contiguous()
        return Linear(wb, bb)

    def pack_linear_f32(self, w, b):
        return LinearF32(self.to_f32(w), None if b is None else self.to_f32(b))

    # ---- GEMM -------------------------------------------------------------------------------------------------
    def linear(self, x, lin, act=None, g1=None, g0=None, res=None, out_f32=False, out=None):
        self._check_dev(x if not isinstance(x, tuple) else x[0], res, out, g1, g0)
        if lin.fp8:
            return self._linear_fp8(x, lin, act, g1, g0, res, out_f32, out)
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1, (x.dtype, x.shape, x.stride())
        M, K = x.shape
        assert K == lin.K, (K, lin.K)
        if out is None:
            out = torch.empty(M, lin.N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=self.device)
        assert out.shape == (M, lin.N) and out.stride(1) == 1
        if res is not None:
            assert res.shape == (M, lin.N) and res.stride(1) == 1
        tok = None if self._timing is None else self._time_begin(
            dict(kind="linear", M=M, N=lin.N, K=K, res=res is not None, out_f32=out.dtype == torch.float32, act=act))
        _check(self.lib.fw_gemm_bf16(
            x.data_ptr(), x.stride(0), lin.w.data_ptr(), lin.w.stride(0), out.data_ptr(), out.stride(0), _dt(out),
            M, lin.N, K, _ptr(lin.b), ACT[act], _ptr(g1), _ptr(g0),
            _ptr(res), 0 if res is None else res.stride(0), FW_DT_NONE if res is None else _dt(res),
            self._stream()), "fw_gemm_bf16")
        self._time_end(tok)
        return out

    def linear_f32(self, x, lin, silu_in=False, act=None):
        assert x.dtype == torch.float32 and x.numel() == lin.K
        out = torch.empty(lin.N, dtype=torch.float32, device=self.device)
        _check(self.lib.fw_gemv_f32(x.data_ptr(), lin.w.data_ptr(), lin.w.stride(0), _ptr(lin.b), out.data_ptr(),
                                    lin.N, lin.K, int(silu_in), ACT[act], self._stream()), "fw_gemv_f32")
        return out

    # ---- norms ------------------------------------------------------------------------------------------------
    def layernorm(self, x, w=None, b=None, scale=None, shift=None, eps=1e-6, out=None):
        assert x.dim() == 2 and x.stride(1) == 1
        self._check_dev(x, w, b, scale, shift, out)
        rows, C = x.shape
        if out is None:
            out = torch.empty(rows, C, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_layernorm_mod(x.data_ptr(), x.stride(0), _dt(x), out.data_ptr(), out.stride(0), rows, C,
                                         _ptr(w), _ptr(b), _ptr(scale), _ptr(shift), float(eps), self._stream()),
               "fw_layernorm_mod")
        return out

    def layernorm_split(self, x, w=None, b=None, scale=None, shift=None, eps=1e-6):
        """LayerNorm (+ affine, + modulation) of an fp32 stream with the bf16 rounding remainder kept: -> (hi, lo), both bf16, hi + lo =
        the fp32 result to ~16 bits (fw_layernorm_mod_split; the head's LayerNorm: a third of the forward's bf16 floor)."""
        assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        self._check_dev(x, w, b, scale, shift)
        rows, C = x.shape
        hi = torch.empty(rows, C, dtype=torch.bfloat16, device=self.device)
        lo = torch.empty(rows, C, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_layernorm_mod_split(x.data_ptr(), x.stride(0), hi.data_ptr(), lo.data_ptr(), hi.stride(0), rows, C,
                                               _ptr(w), _ptr(b), _ptr(scale), _ptr(shift), float(eps), self._stream()),
               "fw_layernorm_mod_split")
        return hi, lo

    def qk_prep(self, x, heads, hd, norm=None, norm_w=None, norm_b=None, eps=1e-6, rope=None, table=None, out_scale=1.0,
                ext_sumsq=None, norm_width=None, out8=None, head_stride8=None):
        """In place on x [rows, heads*hd] (may be a column slice of a wider buffer).  out_scale: multiplied in before the
        single bf16 rounding (the engine folds softmax_scale*log2(e) into q: see attention(q_prescaled=True)).
        ext_sumsq / norm_width (norm="rms_full" only): x is a head slice of a wider row whose sum of squares over all
        `norm_width` channels is supplied per row (tensor parallelism: row_sumsq + all-reduce).
        out8 (uint8 [rows, heads*hd], row-strided ok): x is left untouched and the result goes to out8 as e4m3 bytes
        (fw_qk_prep_fp8: the bits of cast_fp8(qk_prep(x)) without the two extra passes); returns out8.
        head_stride8 > hd: out8 is [rows, heads*head_stride8], every head zero-padded to head_stride8 bytes."""
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == heads * hd
        self._check_dev(x, norm_w, norm_b, table, ext_sumsq, out8)
        tab_rows = 0 if table is None else table.shape[0]
        if table is not None:
            assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[1:] == (hd // 2, 2)
        if out8 is not None:
            hs8 = int(head_stride8 or hd)
            assert out8.dtype == torch.uint8 and out8.shape == (x.shape[0], heads * hs8) and out8.stride(1) == 1
            if ext_sumsq is not None:
                assert norm == "rms_full" and ext_sumsq.dtype == torch.float32 and ext_sumsq.is_contiguous() and ext_sumsq.numel() == x.shape[0]
            _check(self.lib.fw_qk_prep_fp8(x.data_ptr(), x.stride(0), x.shape[0], heads, hd, NORM[norm], _ptr(norm_w), _ptr(norm_b),
                                           float(eps), ROPE[rope], _ptr(table), tab_rows, float(out_scale), _ptr(ext_sumsq),
                                           0 if ext_sumsq is None else int(norm_width), out8.data_ptr(), out8.stride(0), hs8, self._stream()),
                   "fw_qk_prep_fp8")
            return out8
        if ext_sumsq is not None:
            assert norm == "rms_full" and ext_sumsq.dtype == torch.float32 and ext_sumsq.is_contiguous() and ext_sumsq.numel() == x.shape[0]
            _check(self.lib.fw_qk_prep_tp(x.data_ptr(), x.stride(0), x.shape[0], heads, hd, _ptr(norm_w), float(eps), ROPE[rope],
                                          _ptr(table), tab_rows, float(out_scale), ext_sumsq.data_ptr(), int(norm_width),
                                          self._stream()), "fw_qk_prep_tp")
            return x
        _check(self.lib.fw_qk_prep(x.data_ptr(), x.stride(0), x.shape[0], heads, hd, NORM[norm], _ptr(norm_w),
                                   _ptr(norm_b), float(eps), ROPE[rope], _ptr(table), tab_rows, float(out_scale),
                                   self._stream()),
               "fw_qk_prep")
        return x

    # ---- attention ----------------------------------------------------------------------------------------------
    def prepare_v(self, v, heads, hd, batch=1):
        """v [batch*Lk, heads*hd] (strided ok) -> key-permuted transposed copy the attention kernel consumes."""
        assert v.dtype == torch.bfloat16 and v.stride(1) == 1
        Lk = v.shape[0] // batch
        lkp = (Lk + 63) // 64 * 64
        vt = torch.empty(batch, heads, hd, lkp, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_v_transpose(v.data_ptr(), v.stride(0), Lk * v.stride(0), vt.data_ptr(), lkp, batch, heads,
                                       hd, Lk, self._stream()), "fw_v_transpose")
        return vt, Lk

    @staticmethod
    def q_scale(hd):
        """The factor folded into q by qk_prep(out_scale=...) for attention(q_prescaled=True): softmax scale * log2(e)."""
        return 1.4426950408889634 / math.sqrt(hd)

    def attention(self, q, k, v, heads, hd, batch=1, out=None, accumulate=False, v_prepared=None, q_prescaled=False):
        """softmax(q k^T / sqrt(hd)) v per (batch, head); q [batch*Lq, heads*hd], k/v [batch*Lk, heads*hd].
        q_prescaled: q already carries q_scale(hd) (scores are then used directly in the log2 domain)."""
        assert q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and q.stride(1) == 1 and k.stride(1) == 1
        self._check_dev(q, k, v, out)
        Lq = q.shape[0] // batch
        Lk = k.shape[0] // batch
        vt, lk2 = v_prepared if v_prepared is not None else self.prepare_v(v, heads, hd, batch)
        assert lk2 == Lk
        if out is None:
            out = torch.empty(batch * Lq, heads * hd, dtype=torch.bfloat16, device=self.device)
        # scratch for the split-KV route of a tail q-block (PyTorch owns the memory; 0 bytes = the library would not use it)
        ws, ws_bytes = None, 0
        if q_prescaled and self.split_kv:
            ws_bytes = int(self.lib.fw_attention_workspace_bytes(batch, heads, hd, Lq, Lk))
            if ws_bytes:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        tok = None if self._timing is None else self._time_begin(
            dict(kind="attention", hd=hd, Lq=Lq, Lk=Lk, heads=heads, batch=batch))
        _check(self.lib.fw_attention_bf16(
            q.data_ptr(), q.stride(0), Lq * q.stride(0), k.data_ptr(), k.stride(0), Lk * k.stride(0),
            vt.data_ptr(), vt.shape[-1], out.data_ptr(), out.stride(0), Lq * out.stride(0),
            batch, heads, hd, Lq, Lk, 1.0 / math.sqrt(hd), (1 if accumulate else 0) | (2 if q_prescaled else 0),
            _ptr(ws), ws_bytes, self._stream()), "fw_attention_bf16")
        self._time_end(tok)
        return out

    # ---- embeddings / layout ------------------------------------------------------------------------------------
    def sinusoid(self, t, dim):
        """t: 1-element device tensor (bf16 or f32) -> fp32 [dim] (fp64 math on device, no host sync)."""
        t = t.reshape(-1)[:1].contiguous()
        if t.dtype not in (torch.bfloat16, torch.float32):
            t = t.to(torch.float32)
        out = torch.empty(dim, dtype=torch.float32, device=self.device)
        _check(self.lib.fw_sinusoid(t.data_ptr(), _dt(t), out.data_ptr(), dim, self._stream()), "fw_sinusoid")
        return out

    def patchify(self, x, y, kpad):
        """x [1,Cx,F,H2,W2], y [1,Cy,F,H2,W2] or None -> bf16 [L, kpad]."""
        self._check_dev(x, y)
        x = x.contiguous()
        if x.dtype not in (torch.bfloat16, torch.float32):
            x = x.float()
        _, cx, F, H2, W2 = x.shape
        cy = 0
        if y is not None:
            y = y.to(x.dtype).contiguous()
            cy = y.shape[1]
        L = F * (H2 // 2) * (W2 // 2)
        out = torch.empty(L, kpad, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_patchify(x.data_ptr(), cx, _ptr(y), cy, _dt(x), out.data_ptr(), kpad, F, H2, W2,
                                    self._stream()), "fw_patchify")
        return out

    def unpatchify(self, hd_out, F, Hh, Ww, out_dtype):
        assert hd_out.dtype == torch.float32 and hd_out.shape[1] == 64 and hd_out.stride(1) == 1
        out = torch.empty(1, 16, F, 2 * Hh, 2 * Ww, dtype=out_dtype, device=self.device)
        _check(self.lib.fw_unpatchify(hd_out.data_ptr(), hd_out.stride(0), out.data_ptr(), _dt(out), F, Hh, Ww,
                                      self._stream()), "fw_unpatchify")
        return out

    def assemble_tokens(self, patch, special, S, hw):
        """patch bf16 [S*hw, C]; special fp32 [2, n_special, C] -> fp32 [S*(n_special+hw), C]."""
        n_special, C = special.shape[1], special.shape[2]
        out = torch.empty(S * (n_special + hw), C, dtype=torch.float32, device=self.device)
        _check(self.lib.fw_assemble_tokens(patch.data_ptr(), patch.stride(0), special.data_ptr(), out.data_ptr(),
                                           S, hw, n_special, C, self._stream()), "fw_assemble_tokens")
        return out

    def control_patchify(self, ctl):
        """ctl [1, C, F, 16h, 16w] (bf16/f32) -> bf16 [L, C*256]: PixelUnshuffle(8) + k2s2 patch gather of the Wan2.2 control adapter."""
        self._check_dev(ctl)
        ctl = ctl.contiguous()
        if ctl.dtype not in (torch.bfloat16, torch.float32):
            ctl = ctl.float()
        _, C, F, Hp, Wp = ctl.shape
        assert Hp % 16 == 0 and Wp % 16 == 0, ctl.shape
        h, w = Hp // 16, Wp // 16
        out = torch.empty(F * h * w, C * 256, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_control_patchify(ctl.data_ptr(), _dt(ctl), out.data_ptr(), out.stride(0), C, F, h, w, self._stream()),
               "fw_control_patchify")
        return out

    def im2col3x3(self, x, F, h, w):
        """x bf16 [F*h*w, C] token-major -> bf16 [L, 9*C] (3x3, pad 1), column c*9 + ky*3 + kx."""
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] == F * h * w
        C = x.shape[1]
        out = torch.empty(x.shape[0], 9 * C, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_im2col3x3(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), C, F, h, w, self._stream()),
               "fw_im2col3x3")
        return out

    # ---- geometry heads: channels-last feature maps [frames*H*W, C] -------------------------------
    def _bf16_rows(self, x):
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 8 == 0, (x.dtype, x.shape)

    def im2col(self, x, T, H, W, kt, kh, kw, sh=1, sw=1, t0=0, nt=None, relu_in=False, ph=None, pw=None, up=1):
        """Gather of a convolution as GEMM (fw_im2col): [T*H*W, C] -> [nt*Ho*Wo, kt*kh*kw*C], tap-major columns."""
        self._bf16_rows(x)
        assert x.shape[0] == T * H * W
        nt = T - t0 if nt is None else nt
        ph, pw = kh // 2 if ph is None else ph, kw // 2 if pw is None else pw
        C = x.shape[1]
        Ho, Wo = (H * up + 2 * ph - kh) // sh + 1, (W * up + 2 * pw - kw) // sw + 1
        out = torch.empty(nt * Ho * Wo, kt * kh * kw * C, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_im2col(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), C, T, H, W, kt, kh, kw, sh, sw,
                                  ph, pw, up, t0, nt, int(relu_in), self._stream()), "fw_im2col")
        return out

    def conv_gemm(self, x, T, H, W, lin, kt, kh, kw, sh=1, sw=1, t0=0, nt=None, ph=None, pw=None, up=1, act=None, res=None,
                  out_f32=False, out=None):
        """The convolution as ONE implicit-GEMM launch (fw_conv_gemm_bf16): bit-identical to linear(im2col(x, ...), lin, ...)
        without ever writing the gathered matrix.  x [T*H*W, C], C % 64 == 0; lin packed tap-major ([N, kt*kh*kw*C])."""
        self._bf16_rows(x)
        self._check_dev(x, res, out)
        assert x.shape[0] == T * H * W and not lin.fp8
        nt = T - t0 if nt is None else nt
        ph, pw = kh // 2 if ph is None else ph, kw // 2 if pw is None else pw
        C = x.shape[1]
        assert C % 64 == 0 and lin.K == kt * kh * kw * C, (C, lin.K, kt, kh, kw)
        Ho, Wo = (H * up + 2 * ph - kh) // sh + 1, (W * up + 2 * pw - kw) // sw + 1
        M = nt * Ho * Wo
        if out is None:
            out = torch.empty(M, lin.N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=self.device)
        assert out.shape == (M, lin.N) and out.stride(1) == 1
        if res is not None:
            assert res.shape == (M, lin.N) and res.stride(1) == 1
        _check(self.lib.fw_conv_gemm_bf16(
            x.data_ptr(), x.stride(0), C, T, H, W, kt, kh, kw, sh, sw, ph, pw, up, t0, nt,
            lin.w.data_ptr(), lin.w.stride(0), out.data_ptr(), out.stride(0), _dt(out), lin.N,
            _ptr(lin.b), ACT[act], None, None, _ptr(res), 0 if res is None else res.stride(0),
            FW_DT_NONE if res is None else _dt(res), self._stream()), "fw_conv_gemm_bf16")
        return out

    def softmax_rows(self, s, scale, cols_pad):
        assert s.dtype == torch.float32 and s.dim() == 2 and s.stride(1) == 1 and cols_pad % 2 == 0
        out = torch.empty(s.shape[0], cols_pad, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_softmax_rows(s.data_ptr(), s.stride(0), out.data_ptr(), out.stride(0), s.shape[0], s.shape[1], cols_pad,
                                        float(scale), self._stream()), "fw_softmax_rows")
        return out

    # ---- camera pose encoder ---------------------------------------------------------------------------
    def pixel_unshuffle_rows(self, x, r):
        assert x.dim() == 4 and x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32), (x.shape, x.dtype)
        self._check_dev(x)
        Fr, H, W, C = x.shape
        out = torch.empty(Fr * (H // r) * (W // r), C * r * r, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_pixel_unshuffle(x.data_ptr(), _dt(x), out.data_ptr(), out.stride(0), Fr, H, W, C, r, self._stream()),
               "fw_pixel_unshuffle")
        return out

    def group_norm_rows(self, x, frames, groups, w, b, eps=1e-5, relu=False):
        self._bf16_rows(x)
        rows, C = x.shape
        assert rows % frames == 0 and w.dtype == torch.float32 and b.dtype == torch.float32 and w.numel() == C
        out = torch.empty_like(x)
        _check(self.lib.fw_group_norm_rows(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), frames, rows // frames, C,
                                           groups, w.data_ptr(), b.data_ptr(), float(eps), int(relu), self._stream()),
               "fw_group_norm_rows")
        return out

    def time_avg_pool(self, x, frames, hw):
        self._bf16_rows(x)
        assert x.shape[0] == frames * hw
        fout = 1 + (frames - 1) // 2 if frames % 2 else frames // 2
        out = torch.empty(fout * hw, x.shape[1], dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_time_avg_pool(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), frames, hw, x.shape[1],
                                         self._stream()), "fw_time_avg_pool")
        return out, fout

    def activation(self, x, act):
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        out = torch.empty_like(x)
        _check(self.lib.fw_activation(x.data_ptr(), out.data_ptr(), x.numel(), ACT[act], self._stream()), "fw_activation")
        return out

    def resize_bilinear(self, x, N, h, w, H, W):
        self._bf16_rows(x)
        assert x.shape[0] == N * h * w
        out = torch.empty(N * H * W, x.shape[1], dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_resize_bilinear(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), N, h, w, H, W,
                                           x.shape[1], self._stream()), "fw_resize_bilinear")
        return out

    def chan_rmsnorm_silu(self, x, gamma, c_true, silu=True):
        self._bf16_rows(x)
        assert gamma.dtype == torch.float32 and gamma.numel() == x.shape[1]
        out = torch.empty_like(x)
        _check(self.lib.fw_chan_rmsnorm_silu(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
                                             int(c_true), gamma.data_ptr(), int(silu), self._stream()), "fw_chan_rmsnorm_silu")
        return out

    def depth_to_space(self, y, N, h, w, k, C):
        self._bf16_rows(y)
        assert y.shape == (N * h * w, k * k * C)
        out = torch.empty(N * h * k * w * k, C, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_depth_to_space(y.data_ptr(), y.stride(0), out.data_ptr(), out.stride(0), N, h, w, k, C,
                                          self._stream()), "fw_depth_to_space")
        return out

    def add_table(self, x, table):
        self._bf16_rows(x)
        assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[1] == x.shape[1]
        _check(self.lib.fw_add_table(x.data_ptr(), x.stride(0), table.data_ptr(), x.shape[0], table.shape[0], x.shape[1],
                                     self._stream()), "fw_add_table")
        return x

    def unfold_time2(self, y, n, hw, C):
        self._bf16_rows(y)
        assert y.shape == (n * hw, 2 * C)
        out = torch.empty(2 * n * hw, C, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_unfold_time2(y.data_ptr(), y.stride(0), out.data_ptr(), out.stride(0), n, hw, C, self._stream()),
               "fw_unfold_time2")
        return out

    def add_act(self, a, b=None, relu=False):
        assert a.dtype == torch.bfloat16 and a.is_contiguous() and (b is None or (b.shape == a.shape and b.is_contiguous()
                                                                                  and b.dtype == torch.bfloat16))
        out = torch.empty_like(a)
        _check(self.lib.fw_add_act(a.data_ptr(), _ptr(b), out.data_ptr(), a.numel(), int(relu), self._stream()), "fw_add_act")
        return out

    def adaln_rows(self, x, mod):
        assert x.dtype == torch.float32 and mod.dtype == torch.float32 and x.is_contiguous() and mod.is_contiguous()
        rows, C = x.shape
        assert mod.shape == (rows, 3 * C)
        out = torch.empty_like(x)
        _check(self.lib.fw_adaln_rows(x.data_ptr(), mod.data_ptr(), out.data_ptr(), rows, C, 1e-6, self._stream()),
               "fw_adaln_rows")
        return out

    HEAD_MODES = {"exp": 0, "inv_log": 1, "pose": 2}

    def head_activation(self, y, mode):
        assert y.dtype == torch.float32 and y.dim() == 2 and y.is_contiguous()
        rows, n = y.shape
        if mode == "pose":
            out = torch.empty_like(y)
            _check(self.lib.fw_head_activation(y.data_ptr(), rows, n, 2, out.data_ptr(), None, self._stream()), "fw_head_activation")
            return out
        pts = torch.empty(rows, n - 1, dtype=torch.float32, device=self.device)
        conf = torch.empty(rows, dtype=torch.float32, device=self.device)
        _check(self.lib.fw_head_activation(y.data_ptr(), rows, n, self.HEAD_MODES[mode], pts.data_ptr(), conf.data_ptr(),
                                           self._stream()), "fw_head_activation")
        return pts, conf

    # ---- fp8 linear (AutoWrappedLinear.fp8_linear, diffsynth_wan22/vram_management/layers.py:115-151) ----
    def pack_linear_fp8(self, w, b, bias_through_fp8=True):
        """w [N, K] (K % 64 == 0) -> e4m3 bytes (raw cast, scale 1: layers.py:134,137); bias rounded to bf16 (layers.py:138).
        bias_through_fp8: AutoWrappedLinear.forward casts weight AND bias to the computation dtype before calling fp8_linear
        (layers.py:158-159), so in the module the bias passes through e4m3 first; False = fp8_linear called directly."""
        assert w.shape[1] % 64 == 0, w.shape
        wb = w.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
        wq = torch.empty(wb.shape, dtype=torch.uint8, device=self.device)
        _check(self.lib.fw_fp8_quant_rows(wb.data_ptr(), wb.stride(0), wq.data_ptr(), wq.stride(0), None, wb.shape[0], wb.shape[1],
                                          1, self._stream()), "fw_fp8_quant_rows")
        return Linear(wq, None if b is None else self.fp8_bias(b, bias_through_fp8), fp8=True)

    def fp8_bias(self, b, bias_through_fp8=True):
        """The bias values the fp8 linear adds, as fp32 [N]: rounded to bf16 (layers.py:138) and, inside the module, through e4m3 first
        (layers.py:158-159).  Also what a row-parallel fp8 linear adds after its all-reduce (tensor_parallel.py)."""
        bb = b.detach().to(device="cpu", dtype=torch.bfloat16)
        if bias_through_fp8:
            bb = bb.to(torch.float8_e4m3fn).to(torch.bfloat16)           # pack time, [N] values: on the host
        return bb.to(device=self.device, dtype=torch.float32).contiguous()

    def quantize_fp8_rows(self, x, amax=None):
        """x bf16 [M, K] -> (e4m3 bytes [M, K], fp32 scale [M]): the per-row activation quantiser of fp8_linear.
        amax (fp32 [M]): the row maximum is SUPPLIED -- x is a K-slice of a wider row (row-parallel linear under tensor parallelism:
        row_absmax + all-reduce MAX), so every rank divides by the scale of the full row."""
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 8 == 0
        self._check_dev(x, amax)
        q = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        scale = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        if amax is not None:
            assert amax.dtype == torch.float32 and amax.is_contiguous() and amax.numel() == x.shape[0]
            _check(self.lib.fw_fp8_quant_rows_amax(x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), amax.data_ptr(), scale.data_ptr(),
                                                   x.shape[0], x.shape[1], self._stream()), "fw_fp8_quant_rows_amax")
            return q, scale
        _check(self.lib.fw_fp8_quant_rows(x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(), x.shape[0],
                                          x.shape[1], 0, self._stream()), "fw_fp8_quant_rows")
        return q, scale

    def row_absmax(self, x, out=None):
        """x bf16 [rows, w] (column slice ok) -> fp32 [rows] max |x| (exact)."""
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
        self._check_dev(x, out)
        if out is None:
            out = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        _check(self.lib.fw_row_absmax(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), self._stream()), "fw_row_absmax")
        return out

    def _linear_fp8(self, x, lin, act=None, g1=None, g0=None, res=None, out_f32=False, out=None):
        """fp8_linear(x, w, b) with the bf16 linear's fused epilogue: quantise the rows of x, e4m3 x e4m3 GEMM with fp32
        accumulation, * scale_a + bias -> act -> per-column affine -> + residual."""
        # x: bf16 rows, or rows already quantised by quantize_fp8_rows: (e4m3 bytes [M, K], fp32 scale [M])
        q, scale = x if isinstance(x, tuple) else self.quantize_fp8_rows(x)
        M, K = q.shape
        assert K == lin.K and q.dtype == torch.uint8 and q.stride(1) == 1 and scale.numel() == M, (q.shape, lin.K)
        if out is None:
            out = torch.empty(M, lin.N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=self.device)
        assert out.shape == (M, lin.N) and out.stride(1) == 1
        if res is not None:
            assert res.shape == (M, lin.N) and res.stride(1) == 1
        tok = None if self._timing is None else self._time_begin(
            dict(kind="linear", M=M, N=lin.N, K=K, res=res is not None, out_f32=out.dtype == torch.float32, act=act, fp8=True))
        _check(self.lib.fw_gemm_fp8(
            q.data_ptr(), q.stride(0), lin.w.data_ptr(), lin.w.stride(0), scale.data_ptr(), out.data_ptr(), out.stride(0),
            _dt(out), M, lin.N, K, _ptr(lin.b), ACT[act], _ptr(g1), _ptr(g0),
            _ptr(res), 0 if res is None else res.stride(0), FW_DT_NONE if res is None else _dt(res),
            self._stream()), "fw_gemm_fp8")
        self._time_end(tok)
        return out

    # ---- fp8 attention (hd 128; parity unpinned: no reference semantics, see include/fw_mi355x.h) ---------------------------
    FP8_Q_EXP = 3      # Q8 holds q * softmax_scale * log2(e) * 2^3

    def q_scale_fp8(self, hd):
        """out_scale for qk_prep when q is to be cast to e4m3 for attention_fp8."""
        return self.q_scale(hd) * float(2 ** self.FP8_Q_EXP)

    def cast_fp8(self, x, out=None):
        """bf16 [rows, C] (strided rows ok) -> e4m3 bytes, raw cast (round to nearest even, no scale); out: uint8 view to fill."""
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
        self._check_dev(x, out)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        assert out.dtype == torch.uint8 and out.shape == x.shape and out.stride(1) == 1
        _check(self.lib.fw_fp8_quant_rows(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), None, x.shape[0], x.shape[1], 1,
                                          self._stream()), "fw_fp8_quant_rows")
        return out

    def prepare_v_fp8(self, v, heads, hd, batch=1, hd_out=None):
        """v [batch*Lk, heads*hd]: bf16, or e4m3 bytes already (uint8: what the head exchange delivered) -> (Vt8, Lk).
        hd_out > hd: Vt8 has hd_out rows per head, the extra rows zero (a head_dim-96 V for the head_dim-128 kernel)."""
        assert v.dtype in (torch.bfloat16, torch.uint8) and v.stride(1) == 1
        self._check_dev(v)
        Lk = v.shape[0] // batch
        lkp = (Lk + 63) // 64 * 64
        ho = int(hd_out or hd)
        vt = torch.empty(batch, heads, ho, lkp, dtype=torch.uint8, device=self.device)
        if v.dtype == torch.uint8:
            _check(self.lib.fw_v_transpose_e4m3(v.data_ptr(), v.stride(0), Lk * v.stride(0), vt.data_ptr(), lkp, batch, heads, hd, Lk, ho,
                                                self._stream()), "fw_v_transpose_e4m3")
            return vt, Lk
        _check(self.lib.fw_v_transpose_fp8(v.data_ptr(), v.stride(0), Lk * v.stride(0), vt.data_ptr(), lkp, batch, heads, hd, Lk, ho,
                                           self._stream()), "fw_v_transpose_fp8")
        return vt, Lk

    def attention_fp8(self, q8, k8, vt8, heads, hd, Lk, batch=1, out=None):
        """q8 / k8: e4m3 bytes [batch*L, heads*hd] (q pre-multiplied by q_scale_fp8(hd) before the cast), vt8 from prepare_v_fp8."""
        assert q8.dtype == torch.uint8 and k8.dtype == torch.uint8 and q8.stride(1) == 1 and k8.stride(1) == 1
        self._check_dev(q8, k8, vt8, out)
        Lq = q8.shape[0] // batch
        assert k8.shape[0] // batch == Lk
        if out is None:
            out = torch.empty(batch * Lq, heads * hd, dtype=torch.bfloat16, device=self.device)
        tok = None if self._timing is None else self._time_begin(dict(kind="attention", hd=hd, Lq=Lq, Lk=Lk, heads=heads, batch=batch, fp8=True))
        _check(self.lib.fw_attention_fp8(q8.data_ptr(), q8.stride(0), Lq * q8.stride(0), k8.data_ptr(), k8.stride(0), Lk * k8.stride(0),
                                         vt8.data_ptr(), vt8.shape[-1], out.data_ptr(), out.stride(0), Lq * out.stride(0),
                                         batch, heads, hd, Lq, Lk, self.FP8_Q_EXP, self._stream()), "fw_attention_fp8")
        self._time_end(tok)
        return out

    def linear_fp8(self, x, lin, out_f32=False):
        """fp8_linear(x, w, b): (xq wq^T) * scale_a + bias -> x.dtype (or fp32)."""
        assert lin.fp8
        return self._linear_fp8(x, lin, out_f32=out_f32)

    # ---- tensor-parallel helpers (fantasy_world_amd/tensor_parallel.py) and the sampler step ------------------------------------
    def row_sumsq(self, x, out=None):
        """x bf16 [rows, w] (column slice ok) -> fp32 [rows] sum of squares."""
        assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
        self._check_dev(x, out)
        if out is None:
            out = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        _check(self.lib.fw_row_sumsq(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), self._stream()), "fw_row_sumsq")
        return out

    def residual_add(self, x, y, bias=None, g1=None, g0=None):
        """x (fp32 stream, in place) += (y + bias) * g1 + g0; y bf16 / fp32 [rows, C]."""
        assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and y.shape == x.shape and y.stride(1) == 1
        self._check_dev(x, y, bias, g1, g0)
        _check(self.lib.fw_residual_add(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), _dt(y), x.shape[0], x.shape[1],
                                        _ptr(bias), _ptr(g1), _ptr(g0), self._stream()), "fw_residual_add")
        return x

    def cfg_euler_step(self, pos, neg, latents, cfg_scale, dsigma, out=None, dev_params=None):
        """latents + (neg + cfg_scale * (pos - neg)) * dsigma in one launch, bit-identical to the reference's tensor ops
        (model_wan21.py:318-321, flow_match.py:43-53).  dev_params: device fp32 [2] = (cfg_scale, dsigma) read by the kernel."""
        assert pos.dtype == neg.dtype == latents.dtype and pos.shape == neg.shape == latents.shape
        self._check_dev(pos, neg, latents, out, dev_params)
        pos, neg, latents = pos.contiguous(), neg.contiguous(), latents.contiguous()
        if out is None:
            out = torch.empty_like(latents)
        assert out.is_contiguous() and out.dtype == latents.dtype
        _check(self.lib.fw_cfg_euler_step(pos.data_ptr(), neg.data_ptr(), latents.data_ptr(), out.data_ptr(), latents.numel(),
                                          _dt(latents), float(cfg_scale), float(dsigma), _ptr(dev_params), self._stream()),
               "fw_cfg_euler_step")
        return out

    def modulation_tables(self, mod, t, ls2=None):
        """mod fp32 [nblk, rows, C] + t fp32 [t_rows, C] (row r uses t[r % t_rows]) -> table [nblk, rows, C]; with ls2 [nblk, C]
        (rows == 6) also the fc2 epilogue's per-column scale / offset of every VGGT block: (table, g1, g0).  One launch per forward
        for all blocks of a kind (fw_modulation_tables)."""
        assert mod.dtype == torch.float32 and mod.dim() == 3 and mod.is_contiguous() and t.dtype == torch.float32 and t.is_contiguous()
        self._check_dev(mod, t, ls2)
        nblk, rows, C = mod.shape
        t2 = t.reshape(-1, C)
        table = torch.empty_like(mod)
        g1 = g0 = None
        if ls2 is not None:
            assert ls2.dtype == torch.float32 and ls2.shape == (nblk, C) and ls2.is_contiguous() and rows == 6
            g1, g0 = torch.empty_like(ls2), torch.empty_like(ls2)
        _check(self.lib.fw_modulation_tables(mod.data_ptr(), t2.data_ptr(), t2.shape[0], _ptr(ls2), table.data_ptr(), _ptr(g1), _ptr(g0),
                                             nblk, rows, C, self._stream()), "fw_modulation_tables")
        return (table, g1, g0) if ls2 is not None else table

    def cast_act(self, x):
        assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=self.device)
        _check(self.lib.fw_cast_f32_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0],
                                         x.shape[1], self._stream()), "fw_cast_f32_bf16")
        return out

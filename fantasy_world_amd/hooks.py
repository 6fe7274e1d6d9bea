"""Op-level hooks of the reference (boundary B3): the finest-grained surfaces the reference itself exposes for
swapping one op inside an otherwise-reference forward.  Each of them lands on the same C-ABI kernels the engine uses; none of
them is needed when `install()` (boundary B1) replaces the whole joint_forward -- they exist to A/B a single kernel against the
reference's own code, and for callers that only want one piece.

  install_flash_attention      `flash_attention(q, k, v, num_heads, compatibility_mode)`   wan_video_dit.py:28-66   (install.py)
  install_bicross_attention    `BiMultiHeadAttention.attn_implementation` dispatch          fusion/layer/block.py:323-325,393-410
  install_layernorm_kernel     `get_layernorm(hidden, eps, affine, use_kernel=True)`        fusion/layer/block.py:693-708
  HipLinear                    a target for `enable_vram_management(model, module_map={nn.Linear: HipLinear}, ...)`
                               diffsynth_wan21/vram_management/layers.py:145-166 -- the mechanism by which the reference itself
                               swaps nn.Linear for AutoWrappedLinear (and, with computation_dtype float8_e4m3fn, for the fp8 linear
                               diffsynth_wan22/vram_management/layers.py:113-151)
"""
import sys
import types

import torch


def _default_ops(ops, device):
    if ops is None:
        from .hip_ops import HipOps
        ops = HipOps(device or "cuda")
    return ops


# ------------------------------------------------------------------------------------------------ bicross attention
def install_bicross_attention(root, ops=None, device=None):
    """Every `BiMultiHeadAttention` under `root` (an nn.Module, e.g. the fusion model or one IRGBlock) keeps its own projections
    and RoPE (`forward_sdpa`, block.py:532-560) but runs its two attention directions -- the reference's two
    F.scaled_dot_product_attention calls, block.py:598-605 -- through fw_attention_bf16.  Implemented the way the reference
    selects implementations: `attn_implementation` (block.py:393) is set to 'sdpa' and the instance's `forward_sdpa` is rebound.
    Inference only (no masks, no dropout: what joint_forward uses).  Returns an `undo()` callable."""
    ops = _default_ops(ops, device)
    touched = []

    def forward_sdpa(self, x1, x2, attention_mask_1=None, attention_mask_2=None, freqs=None, freqs_dit=None, freqs_agg=None):
        if attention_mask_1 is not None or attention_mask_2 is not None or self.training:
            return type(self).forward_sdpa(self, x1, x2, attention_mask_1, attention_mask_2, freqs, freqs_dit, freqs_agg)
        bsz, L1, _ = x1.shape
        L2 = x2.shape[1]
        q, k = self.m1_proj(x1), self.m2_proj(x2)
        if freqs_dit is not None:
            rope_apply = sys.modules[type(self).__module__].rope_apply
            q = rope_apply(q, freqs=freqs_dit, num_heads=self.num_heads)
            k = rope_apply(k, freqs=freqs_agg, num_heads=self.num_heads)
        v1, v2 = self.values_m1_proj(x1), self.values_m2_proj(x2)
        H, hd, E = self.num_heads, self.head_dim, self.embed_dim
        rows = lambda t, n: ops.to_act(t.reshape(bsz * n, E))
        # softmax scale * log2(e) is folded into q ONCE: direction 1 uses q as queries, direction 2 uses it as keys
        qs = ops.qk_prep(rows(q, L1).clone(), H, hd, out_scale=ops.q_scale(hd))
        kr, v1r, v2r = rows(k, L2), rows(v1, L1), rows(v2, L2)
        o1 = ops.attention(qs, kr, v2r, H, hd, batch=bsz, q_prescaled=True)             # softmax(q k^T) v2
        o2 = ops.attention(kr, qs, v1r, H, hd, batch=bsz, q_prescaled=True)             # softmax(k q^T) v1
        o1 = self.out_m1_proj(o1.view(bsz, L1, E).to(x1.dtype))
        o2 = self.out_m2_proj(o2.view(bsz, L2, E).to(x2.dtype))
        return o1, o2

    for m in root.modules():
        if type(m).__name__ == "BiMultiHeadAttention":
            touched.append((m, getattr(m, "attn_implementation", "eager"), m.__dict__.get("forward_sdpa")))
            m.attn_implementation = "sdpa"
            m.forward_sdpa = types.MethodType(forward_sdpa, m)

    def undo():
        for m, impl, fwd in touched:
            m.attn_implementation = impl
            if fwd is None:
                m.__dict__.pop("forward_sdpa", None)
            else:
                m.forward_sdpa = fwd
    return undo


# ------------------------------------------------------------------------------------------------ LayerNorm kernel
class HipLayerNorm(torch.nn.Module):
    """What `get_layernorm(..., use_kernel=True)` returns after install_layernorm_kernel(): LayerNorm over the last dimension
    (optionally affine) through fw_layernorm_mod; statistics in fp32.  Output in the input's dtype."""

    def __init__(self, hidden_size, eps=1e-6, elementwise_affine=True, ops=None):
        super().__init__()
        self.hidden_size, self.eps = int(hidden_size), float(eps)
        if elementwise_affine:
            self.weight = torch.nn.Parameter(torch.ones(self.hidden_size))
            self.bias = torch.nn.Parameter(torch.zeros(self.hidden_size))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self._ops = ops

    def forward(self, x):
        ops = self._ops if self._ops is not None else _default_ops(None, x.device)
        self._ops = ops
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        x2 = x2 if x2.dtype in (torch.float32, torch.bfloat16) else x2.float()
        x2 = x2.to(ops.device).contiguous()
        w = None if self.weight is None else ops.to_f32(self.weight)
        b = None if self.bias is None else ops.to_f32(self.bias)
        return ops.layernorm(x2, w=w, b=b, eps=self.eps).view(shape).to(x.dtype)


def install_layernorm_kernel(block_module, ops=None, device=None):
    """Rebind the module-level factory `get_layernorm` of FantasyWorld.fusion.layer.block (block.py:693-708) so that
    `use_kernel=True` (CrossModalityBiAttentionBlock(enable_layernorm_kernel=True), block.py:164-167) builds a HipLayerNorm
    instead of apex's FusedLayerNorm (absent on ROCm images).  Must be installed before the blocks are constructed, like
    apex itself.  Returns an `undo()` callable."""
    original = block_module.get_layernorm

    def get_layernorm(hidden_size, eps, affine, use_kernel):
        if use_kernel:
            return HipLayerNorm(hidden_size, eps=eps, elementwise_affine=affine, ops=ops)
        return original(hidden_size, eps, affine, use_kernel)

    block_module.get_layernorm = get_layernorm

    def undo():
        block_module.get_layernorm = original
    return undo


# ------------------------------------------------------------------------------------------------ nn.Linear module map
class HipLinear(torch.nn.Module):
    """Target class for the reference's module swap:

        enable_vram_management(dit, module_map={torch.nn.Linear: HipLinear},
                               module_config=dict(offload_dtype=..., offload_device=..., onload_dtype=..., onload_device=...,
                                                  computation_dtype=torch.bfloat16, computation_device="cuda"))

    (diffsynth_wan21/vram_management/layers.py:145-166 builds `target_module(module, **module_config, vram_limit=..., name=...)`).
    The weights are packed from the wrapped nn.Linear on first use (after checkpoint loading / LoRA merging) and the forward is
    fw_gemm_bf16; with computation_dtype torch.float8_e4m3fn it is the reference's fp8 linear (AutoWrappedLinear.fp8_linear,
    diffsynth_wan22/vram_management/layers.py:115-151) through fw_fp8_quant_rows + fw_gemm_fp8.  No offloading: 288 GB of HBM
    hold the whole model, so the offload / onload arguments are accepted and ignored."""

    def __init__(self, module, offload_dtype=None, offload_device=None, onload_dtype=None, onload_device=None,
                 computation_dtype=torch.bfloat16, computation_device="cuda", vram_limit=None, name="", ops=None, **kwargs):
        super().__init__()
        self.in_features, self.out_features = module.in_features, module.out_features
        self.weight, self.bias = module.weight, module.bias
        self.computation_dtype, self.name = computation_dtype, name
        self.enable_fp8 = computation_dtype in (getattr(torch, "float8_e4m3fn", None),)
        dev = computation_device if str(computation_device).startswith("cuda") else "cuda"
        self._ops, self._device, self._packed = ops, dev, None

    def _pack(self):
        ops = self._ops = _default_ops(self._ops, self._device)
        w = self.weight.detach().float()
        k_pad = (w.shape[1] + 63) // 64 * 64
        if k_pad != w.shape[1]:
            w = torch.cat([w, w.new_zeros(w.shape[0], k_pad - w.shape[1])], dim=1)
        self._packed = (ops.pack_linear(w, None if self.bias is None else self.bias.detach().float(), fp8=self.enable_fp8),
                        self.weight._version, k_pad)

    def forward(self, x, *args, **kwargs):
        if self._packed is None or self._packed[1] != self.weight._version:      # re-pack after an in-place weight edit (LoRA merge)
            self._pack()
        lin, _, k_pad = self._packed
        ops = self._ops
        shape = x.shape
        x2 = ops.to_act(x.reshape(-1, shape[-1]))
        if k_pad != shape[-1]:
            x2 = torch.cat([x2, x2.new_zeros(x2.shape[0], k_pad - shape[-1])], dim=1)
        return ops.linear(x2, lin).view(*shape[:-1], self.out_features).to(x.dtype)

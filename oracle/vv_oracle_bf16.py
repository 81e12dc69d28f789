"""bf16 YARDSTICK of the acoustic model -- TEST INFRASTRUCTURE, NOT PRODUCT (same rules as vv_oracle.py: only tests/,
smoke() and bench.py's cpu_baseline leg may import it).

The float64 oracle (vv_oracle.Oracle, dtype=torch.float64) with ONE change: a value is rounded to bf16 (nearest even)
wherever the bf16 model keeps it in bf16 -- matrix-pipe operands and activations stored between kernels -- and nowhere
else.  Every sum, normalisation, transcendental and the residual stream stay float64, so what this run differs from the
float64 run by is the cost of the bf16 STORAGE FORMAT alone, independent of any kernel of the product: it is the
yardstick the HIP bf16 path is held to (tests/test_fullsize31_gpu.py), instead of figures the same kernels produced.

Rounding points (the data flow of vietvoice-tts_amd/csrc/vv_api.hip, transformer_impl / preprocess_impl):
  weights        every GEMM / pos-conv weight of the acoustic model and of the text blocks' pointwise convs; biases,
                 embedding, depthwise convs, norm parameters, the time MLP and the AdaLN projections stay fp32
  text blocks    LayerNorm output, GELU(pwconv1) output, GRN output (in place); the block's residual stream stays fp32
  input embed    [x | cond] rows packed for the input projection; its output h; pos-conv 1 output; pos-conv 2 + h -> fp32
  DiT block      modulated LayerNorm outputs; roped q and k and v as the QKV GEMM stores them; q again after the softmax
                 scale * log2(e) is folded in (the attention kernel converts Q once more); the probabilities P as the PV
                 operand (the row sum l adds the UNROUNDED p); the attention output; the gated deltas gate * (out + bias)
                 of both branches; GELU(ff1) output.  The residual x + d_attn + d_mlp is fp32 (here float64).
  head           final modulated LayerNorm output; the flow prediction, CFG combine and Euler update are fp32.
PARITY UNPINNED against the real reference graphs, like the oracle it derives from.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .vv_oracle import Oracle


def rb(x: torch.Tensor) -> torch.Tensor:
    """Round to the bf16 grid (via fp32, as the kernels' fp32 accumulators are), keep the working dtype."""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


_BF16_WEIGHTS = (".pwconv1.weight", ".pwconv2.weight", "input.proj.weight", "input.pos_conv1.weight", "input.pos_conv2.weight",
                 ".attn.qkv.weight", ".attn.out.weight", ".ff1.weight", ".ff2.weight", "final.proj.weight")


class Bf16Oracle(Oracle):
    def __init__(self, spec, weights: Dict[str, torch.Tensor], nfe_step: int = 32):
        super().__init__(spec, weights, nfe_step=nfe_step, dtype=torch.float64)
        for k in list(self.w):
            if k.endswith(_BF16_WEIGHTS) and not k.startswith("voc."):
                self.w[k] = rb(self.w[k])

    # ------------------------------------------------------------------ preprocess: the text blocks' pointwise convs
    def text_embed(self, text_ids: torch.Tensor, n: int, drop: bool) -> torch.Tensor:
        s, w = self.spec, self.w
        ids = text_ids.to(torch.long) + 1
        ids = ids[:n]
        ids = F.pad(ids, (0, n - ids.shape[0]), value=0)
        if drop:
            ids = torch.zeros_like(ids)
        x = w["text.embed.weight"][ids] + self.text_pos_table(n)
        for i in range(s.text_layers):
            p = f"text.blocks.{i}"
            h = F.conv1d(x.t().unsqueeze(0), w[p + ".dwconv.weight"], w[p + ".dwconv.bias"],
                         padding=s.text_conv_k // 2, groups=s.text_dim).squeeze(0).t()
            h = rb(F.layer_norm(h, (s.text_dim,), w[p + ".norm.weight"], w[p + ".norm.bias"], eps=1e-6))
            h = rb(F.gelu(F.linear(h, w[p + ".pwconv1.weight"], w[p + ".pwconv1.bias"])))
            h = rb(self.grn(h, w[p + ".grn.gamma"], w[p + ".grn.beta"]))
            x = x + F.linear(h, w[p + ".pwconv2.weight"], w[p + ".pwconv2.bias"])
        return x

    # ------------------------------------------------------------------ transformer
    def input_embed(self, x: torch.Tensor, cat: torch.Tensor) -> torch.Tensor:
        s, w = self.spec, self.w
        h = rb(F.linear(rb(torch.cat([x, cat], dim=-1)), w["input.proj.weight"], w["input.proj.bias"]))
        pad = s.pos_conv_k // 2
        c = rb(F.mish(F.conv1d(h.t().unsqueeze(0), w["input.pos_conv1.weight"], w["input.pos_conv1.bias"], padding=pad, groups=s.pos_conv_groups)))
        c = F.mish(F.conv1d(c, w["input.pos_conv2.weight"], w["input.pos_conv2.bias"], padding=pad, groups=s.pos_conv_groups))
        return c.squeeze(0).t() + h

    def attention_core(self, a: torch.Tensor, p: str, ropes) -> torch.Tensor:
        """Modulated, bf16 LayerNorm output a (n, D) -> bf16 attention output (n, D) (before the output projection)."""
        s, w = self.spec, self.w
        n = a.shape[0]
        qkv = F.linear(a, w[p + ".attn.qkv.weight"], w[p + ".attn.qkv.bias"])
        q, k, v = qkv.split(s.dim, dim=-1)
        # the QKV epilogue ropes q and k with the UNSCALED angles and stores bf16; attention folds scale * log2(e) into Q and rounds again
        q = rb(self.rope_apply(q.reshape(n, s.heads, s.head_dim), ropes[2], ropes[3]))
        k = rb(self.rope_apply(k.reshape(n, s.heads, s.head_dim), ropes[2], ropes[3]))
        v = rb(v.reshape(n, s.heads, s.head_dim))
        q = rb(q * (float(s.head_dim) ** -0.5 * math.log2(math.e)))
        sc = torch.einsum("qhd,khd->hqk", q, k)                       # base-2 logits
        pr = torch.exp2(sc - sc.amax(dim=-1, keepdim=True))
        o = torch.einsum("hqk,khd->qhd", rb(pr), v) / pr.sum(dim=-1).t().unsqueeze(-1)
        return rb(o.reshape(n, s.dim))

    def dit_forward(self, x: torch.Tensor, cat: torch.Tensor, ropes, step: int, n_blocks: Optional[int] = None) -> torch.Tensor:
        s, w = self.spec, self.w
        d = s.dim
        temb = F.silu(self.time_embed(step))
        h = self.input_embed(x, cat)
        for i in range(s.depth if n_blocks is None else n_blocks):
            p = f"blocks.{i}"
            mod = F.linear(temb, w[p + ".adaln.weight"], w[p + ".adaln.bias"])
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6)
            a = rb(F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_a) + sh_a)
            o = self.attention_core(a, p, ropes)
            h = h + rb(g_a * F.linear(o, w[p + ".attn.out.weight"], w[p + ".attn.out.bias"]))
            m = rb(F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_m) + sh_m)
            m = rb(F.gelu(F.linear(m, w[p + ".ff1.weight"], w[p + ".ff1.bias"]), approximate="tanh"))
            h = h + rb(g_m * F.linear(m, w[p + ".ff2.weight"], w[p + ".ff2.bias"]))
        mod = F.linear(temb, w["final.adaln.weight"], w["final.adaln.bias"])
        sc_f, sh_f = mod.chunk(2)
        h = rb(F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_f) + sh_f)
        return F.linear(h, w["final.proj.weight"], w["final.proj.bias"])

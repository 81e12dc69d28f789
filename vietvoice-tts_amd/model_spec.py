"""Architecture constants of the synthesis hot path and the seeded synthetic weights.

The reference hides every one of these inside three ONNX graphs that are not in
the repository (reference: vietvoicetts/core/model.py:73-77, model_config.py:26);
only sample_rate=24000 / hop_length=256 / nfe_step=32 are visible
(model_config.py:29-34).  SURVEY.md section 8(a) "Model constants" fixes the
builder-chosen defaults restated here; nothing in the reference contradicts or
confirms them ("parity unpinned" against the real graphs).

Weights are synthetic: there is no network for checkpoints, so a seeded
generator produces variance-scaled tensors in torch-native layouts
(Linear [out,in], Conv1d [out,in/groups,k], ConvTranspose1d [in,out,k]).  The
HIP path and the CPU oracle consume the same dict.
"""
from __future__ import annotations

import json
import math
from dataclasses import asdict, dataclass, field
from typing import Dict, Tuple

import torch


@dataclass(frozen=True)
class ModelSpec:
    # mel front-end (preprocess graph)
    sample_rate: int = 24000
    n_fft: int = 1024
    win_length: int = 1024
    hop_length: int = 256
    n_mel: int = 100
    # acoustic model (transformer graph)
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    head_dim: int = 64
    ff_mult: int = 2
    text_dim: int = 512
    text_layers: int = 4
    text_conv_k: int = 7
    text_ff_mult: int = 2
    vocab_size: int = 256          # text ids are 0..vocab_size-1; embedding has vocab_size+1 rows
    pos_conv_k: int = 31
    pos_conv_groups: int = 16
    time_freq_dim: int = 256
    cfg_strength: float = 2.0
    sway_coef: float = -1.0
    rope_theta: float = 10000.0
    # vocoder (decode graph)
    voc_pre_ch: int = 512
    voc_pre_k: int = 7
    voc_post_k: int = 7
    voc_up_rates: Tuple[int, ...] = (8, 8, 2, 2)
    voc_up_kernels: Tuple[int, ...] = (16, 16, 4, 4)
    voc_res_kernels: Tuple[int, ...] = (3, 7, 11)
    voc_res_dilations: Tuple[int, ...] = (1, 3, 5)
    voc_lrelu: float = 0.1

    def __post_init__(self):
        assert self.heads * self.head_dim == self.dim
        assert self.head_dim == 64, "attention kernels are written for head_dim 64"
        assert self.dim // self.pos_conv_groups == 64, "pos-conv kernels assume 64 channels per group"
        prod = 1
        for r, k in zip(self.voc_up_rates, self.voc_up_kernels):
            assert k == 2 * r and r % 2 == 0
            prod *= r
        assert prod == self.hop_length
        assert self.n_mel == 100 or self.n_mel % 4 == 0

    @property
    def cat_dim(self) -> int:
        return 2 * self.n_mel + self.text_dim

    @property
    def cond_dim(self) -> int:
        return self.n_mel + self.text_dim

    def voc_channels(self):
        ch = [self.voc_pre_ch]
        for _ in self.voc_up_rates:
            ch.append(ch[-1] // 2)
        return ch

    def to_json(self) -> str:
        return json.dumps(asdict(self))

    @classmethod
    def from_json(cls, s: str) -> "ModelSpec":
        d = json.loads(s)
        for k in ("voc_up_rates", "voc_up_kernels", "voc_res_kernels", "voc_res_dilations"):
            d[k] = tuple(d[k])
        return cls(**d)

    @classmethod
    def full(cls) -> "ModelSpec":
        return cls()

    @classmethod
    def tiny(cls) -> "ModelSpec":
        """Small dims for oracle-speed parity tests; same topology, same kernels."""
        return cls(dim=128, depth=2, heads=2, text_dim=128, text_layers=2, vocab_size=64,
                   pos_conv_groups=2, voc_pre_ch=64)

    @classmethod
    def small(cls) -> "ModelSpec":
        """Mid-size config (all tile paths exercised, oracle still seconds)."""
        return cls(dim=256, depth=3, heads=4, text_dim=128, text_layers=2, vocab_size=64,
                   pos_conv_groups=4, voc_pre_ch=128)


def time_grid(nfe_step: int, sway_coef: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """ODE time grid: nfe_step points on [0,1] with sway sampling, and the Euler deltas.

    The reference evaluates the transformer ``len(range(0, nfe_step-1, fuse_nfe))``
    times (core/tts_engine.py:157-159), i.e. nfe_step-1 = 31 Euler steps at defaults.
    Returns (t[:-1], dt) each of length nfe_step-1, float64 computed then cast to fp32.
    """
    t = torch.linspace(0.0, 1.0, nfe_step, dtype=torch.float64)
    t = t + sway_coef * (torch.cos(math.pi / 2 * t) - 1.0 + t)
    dt = t[1:] - t[:-1]
    return t[:-1].to(torch.float32), dt.to(torch.float32)


def mel_filterbank(spec: ModelSpec) -> torch.Tensor:
    """HTK-scale triangular mel filterbank, no norm: (n_fft//2+1, n_mel) fp32."""
    n_freqs = spec.n_fft // 2 + 1
    f_max = spec.sample_rate / 2.0
    all_freqs = torch.linspace(0, spec.sample_rate // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, spec.n_mel + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    return fb.to(torch.float32).contiguous()


def weight_shapes(spec: ModelSpec) -> Dict[str, Tuple[Tuple[int, ...], float]]:
    """name -> (shape, std).  std==0.0 means zeros; std<0 means constant |std| (ones-like)."""
    D, Dt, M = spec.dim, spec.text_dim, spec.n_mel
    sh: Dict[str, Tuple[Tuple[int, ...], float]] = {}

    def lin(name, out_f, in_f, gain=1.0, bias_std=0.02):
        sh[name + ".weight"] = ((out_f, in_f), gain / math.sqrt(in_f))
        sh[name + ".bias"] = ((out_f,), bias_std)

    # ---- text embedding + ConvNeXtV2 blocks
    sh["text.embed.weight"] = ((spec.vocab_size + 1, Dt), 1.0)
    for i in range(spec.text_layers):
        p = f"text.blocks.{i}"
        sh[p + ".dwconv.weight"] = ((Dt, 1, spec.text_conv_k), 1.0 / math.sqrt(spec.text_conv_k))
        sh[p + ".dwconv.bias"] = ((Dt,), 0.02)
        sh[p + ".norm.weight"] = ((Dt,), -1.0)
        sh[p + ".norm.bias"] = ((Dt,), 0.02)
        lin(p + ".pwconv1", Dt * spec.text_ff_mult, Dt, gain=1.4)
        sh[p + ".grn.gamma"] = ((Dt * spec.text_ff_mult,), 0.3)
        sh[p + ".grn.beta"] = ((Dt * spec.text_ff_mult,), 0.02)
        lin(p + ".pwconv2", Dt, Dt * spec.text_ff_mult, gain=0.5)
    # ---- input embedding
    lin("input.proj", D, spec.cat_dim)
    cg = D // spec.pos_conv_groups
    for j in (1, 2):
        sh[f"input.pos_conv{j}.weight"] = ((D, cg, spec.pos_conv_k), 1.0 / math.sqrt(cg * spec.pos_conv_k))
        sh[f"input.pos_conv{j}.bias"] = ((D,), 0.02)
    # ---- time embedding
    lin("time.mlp1", D, spec.time_freq_dim)
    lin("time.mlp2", D, D)
    # ---- DiT blocks
    for i in range(spec.depth):
        p = f"blocks.{i}"
        lin(p + ".adaln", 6 * D, D, gain=0.6, bias_std=0.05)
        lin(p + ".attn.qkv", 3 * D, D)
        lin(p + ".attn.out", D, D, gain=0.7)
        lin(p + ".ff1", D * spec.ff_mult, D, gain=1.4)
        lin(p + ".ff2", D, D * spec.ff_mult, gain=0.7)
    lin("final.adaln", 2 * D, D, gain=0.6, bias_std=0.05)
    lin("final.proj", M, D, gain=1.0)
    # ---- vocoder
    ch = spec.voc_channels()
    sh["voc.pre.weight"] = ((ch[0], M, spec.voc_pre_k), 1.0 / math.sqrt(M * spec.voc_pre_k) / 3.0)
    sh["voc.pre.bias"] = ((ch[0],), 0.02)
    for s, (r, k) in enumerate(zip(spec.voc_up_rates, spec.voc_up_kernels)):
        cin, cout = ch[s], ch[s + 1]
        # each output sample sees 2*cin taps (k = 2*stride)
        sh[f"voc.up.{s}.weight"] = ((cin, cout, k), 1.4 / math.sqrt(2 * cin))
        sh[f"voc.up.{s}.bias"] = ((cout,), 0.02)
        for a, rk in enumerate(spec.voc_res_kernels):
            for b, _d in enumerate(spec.voc_res_dilations):
                q = f"voc.res.{s}.{a}.{b}"
                sh[q + ".conv1.weight"] = ((cout, cout, rk), 1.4 / math.sqrt(cout * rk))
                sh[q + ".conv1.bias"] = ((cout,), 0.02)
                sh[q + ".conv2.weight"] = ((cout, cout, rk), 0.45 / math.sqrt(cout * rk))
                sh[q + ".conv2.bias"] = ((cout,), 0.02)
    sh["voc.post.weight"] = ((1, ch[-1], spec.voc_post_k), 0.35 / math.sqrt(ch[-1] * spec.voc_post_k))
    sh["voc.post.bias"] = ((1,), 0.0)
    return sh


def make_synthetic_weights(spec: ModelSpec, seed: int = 9527) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights (CPU).  One generator per tensor so any subset is reproducible."""
    out: Dict[str, torch.Tensor] = {}
    for idx, (name, (shape, std)) in enumerate(weight_shapes(spec).items()):
        if std < 0:
            g = torch.Generator().manual_seed(seed * 1000003 + idx)
            t = torch.full(shape, -std, dtype=torch.float32) + 0.05 * torch.randn(shape, generator=g)
        elif std == 0.0:
            t = torch.zeros(shape, dtype=torch.float32)
        else:
            g = torch.Generator().manual_seed(seed * 1000003 + idx)
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        out[name] = t
    return out


def count_params(spec: ModelSpec) -> Dict[str, int]:
    tot = {"acoustic": 0, "vocoder": 0}
    for name, (shape, _s) in weight_shapes(spec).items():
        n = 1
        for d in shape:
            n *= d
        tot["vocoder" if name.startswith("voc.") else "acoustic"] += n
    return tot

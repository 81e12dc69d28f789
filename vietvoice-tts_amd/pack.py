"""Device weight layout: torch-native fp32 tensors -> the layouts the gfx950 kernels read.

One flat byte buffer holds every tensor (256-byte aligned slots), so a multi-GPU job moves the
model with ONE RCCL broadcast of that buffer and each rank binds the same
(name, offset) table.  ``plan()`` derives the table from shapes alone; ``fill()`` materialises it.

Layouts:
  * GEMM weights: [N_pad128][K_pad64] row-major (torch Linear order, K contiguous), dtype = acoustic
    dtype (bf16/f32); biases fp32 [N_pad128]; zero padding.
  * time / AdaLN projections: fp32 [N][K] (run once per time grid).
  * pos-conv: bf16 [G][KW][64 co][64 ci];  f32 [G][KW][64 ci][64 co].
  * vocoder conv: fp32 [Cin_pad8][KW][Cout_pad64]; ConvTranspose (stride u, kernel 2u) in polyphase
    form fp32 [Cin_pad8][2][(Cout*u)_pad64] with row = co*u + phase, tap j -> kernel index phase + j*u.
  * constant tables (Hann window, DFT twiddles, mel filterbank, text position table) fp32.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Tuple

import torch

from .model_spec import ModelSpec, mel_filterbank

ALIGN = 256
MAX_POS = 4096


def _pad_to(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def _pad2(t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    out = torch.zeros((rows, cols), dtype=t.dtype)
    out[: t.shape[0], : t.shape[1]] = t
    return out


def _pad1(t: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros((n,), dtype=t.dtype)
    out[: t.shape[0]] = t
    return out


def rope_tables(spec: ModelSpec, max_pos: int = MAX_POS):
    """Tables are DEFINED in float64 (angle = pos * theta^(-2i/hd), then cos/sin, then one rounding to fp32) so that any
    implementation reproduces them to an fp32 ulp; an fp32 angle would carry ~1.5e-4 rad of rounding at position 4096."""
    inv = 1.0 / (spec.rope_theta ** (torch.arange(0, spec.head_dim, 2, dtype=torch.float64) / spec.head_dim))
    ang = torch.outer(torch.arange(max_pos, dtype=torch.float64), inv).repeat_interleave(2, dim=-1)
    scale = spec.head_dim ** -0.5
    f = lambda t: t.to(torch.float32).contiguous()
    return f(ang.cos() * scale), f(ang.sin() * scale), f(ang.cos()), f(ang.sin())


def text_pos_table(spec: ModelSpec, max_pos: int = MAX_POS) -> torch.Tensor:
    d = spec.text_dim
    freqs = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64)[: d // 2] / d))
    ang = torch.outer(torch.arange(max_pos, dtype=torch.float64), freqs)
    return torch.cat([ang.cos(), ang.sin()], dim=-1).to(torch.float32).contiguous()


def time_sinus_table(spec: ModelSpec, t_grid: torch.Tensor) -> torch.Tensor:
    half = spec.time_freq_dim // 2
    emb = torch.exp(torch.arange(half, dtype=torch.float64) * -(math.log(10000.0) / (half - 1)))
    arg = 1000.0 * t_grid.to(torch.float32).to(torch.float64).unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat([arg.sin(), arg.cos()], dim=-1).to(torch.float32).contiguous()


Entry = Tuple[str, torch.dtype, Tuple[int, ...], Callable[[Dict[str, torch.Tensor]], torch.Tensor]]


def entries(spec: ModelSpec, acoustic_dtype: torch.dtype) -> List[Entry]:
    """(name, dtype, shape, producer) for every device tensor, in a fixed order."""
    adt = acoustic_dtype
    assert adt in (torch.bfloat16, torch.float32)
    D, Dt, M = spec.dim, spec.text_dim, spec.n_mel
    C2 = Dt * spec.text_ff_mult
    FF = D * spec.ff_mult
    KP = _pad_to(spec.cat_dim, 64)
    MP = _pad_to(M, 128)
    es: List[Entry] = []

    def add(name, dtype, shape, fn):
        es.append((name, dtype, tuple(shape), fn))

    def lin(name, n, k, np_=None, kp=None, dtype=None):
        np_, kp, dt = np_ or _pad_to(n, 128), kp or _pad_to(k, 64), dtype or adt
        add(name + ".weight", dt, (np_, kp), lambda w, name=name, np_=np_, kp=kp, dt=dt: _pad2(w[name + ".weight"], np_, kp).to(dt))
        add(name + ".bias", torch.float32, (np_,), lambda w, name=name, np_=np_: _pad1(w[name + ".bias"], np_))

    def f32(name, shape, fn):
        add(name, torch.float32, shape, fn)

    # ---- constant tables
    f32("const.window", (spec.n_fft,), lambda w: torch.hann_window(spec.win_length, periodic=True, dtype=torch.float32))
    ang = 2.0 * math.pi * torch.arange(spec.n_fft, dtype=torch.float64) / spec.n_fft
    f32("const.tw_cos", (spec.n_fft,), lambda w: ang.cos().to(torch.float32))
    f32("const.tw_sin", (spec.n_fft,), lambda w: ang.sin().to(torch.float32))
    f32("const.mel_fb", (spec.n_fft // 2 + 1, M), lambda w: mel_filterbank(spec))
    f32("const.text_pos", (MAX_POS, Dt), lambda w: text_pos_table(spec))
    # ---- text
    f32("text.embed.weight", (spec.vocab_size + 1, Dt), lambda w: w["text.embed.weight"])
    for i in range(spec.text_layers):
        p = f"text.blocks.{i}"
        f32(p + ".dwconv.weight", (Dt, spec.text_conv_k), lambda w, p=p: w[p + ".dwconv.weight"].reshape(Dt, spec.text_conv_k))
        f32(p + ".dwconv.bias", (Dt,), lambda w, p=p: w[p + ".dwconv.bias"])
        f32(p + ".norm.weight", (Dt,), lambda w, p=p: w[p + ".norm.weight"])
        f32(p + ".norm.bias", (Dt,), lambda w, p=p: w[p + ".norm.bias"])
        lin(p + ".pwconv1", C2, Dt)
        f32(p + ".grn.gamma", (C2,), lambda w, p=p: w[p + ".grn.gamma"])
        f32(p + ".grn.beta", (C2,), lambda w, p=p: w[p + ".grn.beta"])
        lin(p + ".pwconv2", Dt, C2)
    # ---- input embedding
    lin("input.proj", D, spec.cat_dim, kp=KP)
    G, KW = spec.pos_conv_groups, spec.pos_conv_k
    for j in (1, 2):
        n = f"input.pos_conv{j}"
        if adt == torch.bfloat16:
            add(n + ".weight", adt, (G, KW, 64, 64),
                lambda w, n=n: w[n + ".weight"].reshape(G, 64, 64, KW).permute(0, 3, 1, 2).contiguous().to(adt))
        else:
            add(n + ".weight", adt, (G, KW, 64, 64),
                lambda w, n=n: w[n + ".weight"].reshape(G, 64, 64, KW).permute(0, 3, 2, 1).contiguous())
        f32(n + ".bias", (D,), lambda w, n=n: w[n + ".bias"])
    # ---- time embedding / AdaLN projections (fp32, once per grid)
    lin("time.mlp1", D, spec.time_freq_dim, kp=spec.time_freq_dim, dtype=torch.float32)
    lin("time.mlp2", D, D, kp=D, dtype=torch.float32)
    for i in range(spec.depth):
        p = f"blocks.{i}"
        lin(p + ".adaln", 6 * D, D, kp=D, dtype=torch.float32)
        lin(p + ".attn.qkv", 3 * D, D)
        lin(p + ".attn.out", D, D)
        lin(p + ".ff1", FF, D)
        lin(p + ".ff2", D, FF)
    lin("final.adaln", 2 * D, D, kp=D, dtype=torch.float32)
    lin("final.proj", M, D, np_=MP)
    # ---- vocoder (fp32)
    ch = spec.voc_channels()

    def conv_w(name, cout, cin, kw):
        cp, rp = _pad_to(cin, 8), _pad_to(cout, 64)

        def fn(w, name=name):
            t = torch.zeros((cp, kw, rp), dtype=torch.float32)
            t[:cin, :, :cout] = w[name + ".weight"].permute(1, 2, 0)
            return t
        f32(name + ".weight", (cp, kw, rp), fn)
        f32(name + ".bias", (cout,), lambda w, name=name: w[name + ".bias"])

    conv_w("voc.pre", ch[0], M, spec.voc_pre_k)
    for s, (u, k) in enumerate(zip(spec.voc_up_rates, spec.voc_up_kernels)):
        cin, cout = ch[s], ch[s + 1]
        cp, rp = _pad_to(cin, 8), _pad_to(cout * u, 64)
        n = f"voc.up.{s}"

        def up_fn(w, n=n, cin=cin, cout=cout, u=u, cp=cp, rp=rp):
            t = torch.zeros((cp, 2, rp), dtype=torch.float32)
            # torch ConvTranspose1d weight [cin][cout][2u] -> [cin][j][co*u + p] = W[cin][co][p + j*u]
            t[:cin, :, : cout * u] = w[n + ".weight"].reshape(cin, cout, 2, u).permute(0, 2, 1, 3).reshape(cin, 2, cout * u)
            return t
        f32(n + ".weight", (cp, 2, rp), up_fn)
        f32(n + ".bias", (cout,), lambda w, n=n: w[n + ".bias"])
        for a, rk in enumerate(spec.voc_res_kernels):
            for b, _d in enumerate(spec.voc_res_dilations):
                for c in (1, 2):
                    conv_w(f"voc.res.{s}.{a}.{b}.conv{c}", cout, cout, rk)
    f32("voc.post.weight", (ch[-1], spec.voc_post_k), lambda w: w["voc.post.weight"].reshape(ch[-1], spec.voc_post_k))
    f32("voc.post.bias", (1,), lambda w: w["voc.post.bias"])
    return es


def _nbytes(dtype: torch.dtype, shape) -> int:
    n = 1
    for d in shape:
        n *= d
    return n * (2 if dtype == torch.bfloat16 else 4)


def plan(spec: ModelSpec, acoustic_dtype: torch.dtype):
    """[(name, offset, nbytes)], total_bytes -- shapes only, identical on every rank."""
    table, off = [], 0
    for name, dt, shape, _fn in entries(spec, acoustic_dtype):
        nb = _nbytes(dt, shape)
        table.append((name, off, nb))
        off = _pad_to(off + nb, ALIGN)
    return table, off


def fill(spec: ModelSpec, acoustic_dtype: torch.dtype, weights: Dict[str, torch.Tensor], flat_cpu: torch.Tensor) -> None:
    """Materialise every tensor into the flat uint8 CPU buffer (rank 0 only in a multi-GPU job)."""
    table, total = plan(spec, acoustic_dtype)
    assert flat_cpu.dtype == torch.uint8 and flat_cpu.numel() >= total
    for (name, dt, shape, fn), (_n, off, nb) in zip(entries(spec, acoustic_dtype), table):
        t = fn(weights).to(dt).contiguous()
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), shape)
        flat_cpu[off: off + nb] = t.view(torch.uint8).reshape(-1)

"""CPU fp32 ORACLE of the synthesis hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (vietvoice_tts_amd.*) never does, and fails
loudly when the HIP extension is missing.

PARITY UNPINNED against the real reference graphs: the arithmetic of the
reference hot path lives in preprocess.onnx / transformer.onnx / decode.onnx
(reference vietvoicetts/core/model.py:73-77), executed by onnxruntime
(model.py:98-102; call sites core/tts_engine.py:146,172,187).  Neither the
graphs/weights (download-only, model_config.py:26,71-104) nor onnxruntime are
available offline, and no reference test pins a number produced by them
(tests/test_tts_engine_full.py:54-75 mocks the sessions).  This file is
therefore a plain-torch restatement of the architecture SURVEY.md section 8(a)
fixes (flow-matching DiT acoustic model + transposed-conv/MRF vocoder), and it
is the numerical ground truth for every HIP kernel.  What IS pinned is the I/O
contract of the three graphs as the reference drives them:

  preprocess(audio int16 (1,1,S), text_ids int32 (1,T), max_duration int64 (1,))
      -> noise, rope_cos_q, rope_sin_q, rope_cos_k, rope_sin_k,
         cat_mel_text, cat_mel_text_drop, ref_signal_len      (tts_engine.py:133-146, 229-230)
  transformer(8 inputs) -> (noise, time_step), called nfe_step-1 times (tts_engine.py:148-174)
  decode(noise, ref_signal_len) -> int16 PCM                  (tts_engine.py:176-187)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

import numpy as np

# The oracle imports NOTHING from the product package: the constant tables below (mel filterbank, sway-sampled
# time grid, sinusoidal time table, rope / text position tables, Hann window) are its own float64 closed forms,
# written independently of vietvoice_tts_amd/model_spec.py and pack.py.  tests/test_oracle_tables_cpu.py asserts
# that the two sets agree, so a wrong table on either side is visible.  ``spec`` is duck-typed (any object with
# the architecture constants of SURVEY.md 8(a)); weights are a plain name -> tensor dict handed in by the caller.


def oracle_time_grid(nfe_step: int, sway_coef: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sway-sampled ODE grid t_i = u_i + s*(cos(pi/2 u_i) - 1 + u_i), u_i = i/(nfe-1); the reference runs
    nfe_step-1 evaluations (core/tts_engine.py:157-159).  float64, then fp32."""
    u = np.arange(nfe_step, dtype=np.float64) / float(nfe_step - 1)
    t = u + sway_coef * (np.cos(0.5 * np.pi * u) - 1.0 + u)
    return torch.from_numpy(t[:-1].astype(np.float32)), torch.from_numpy(np.diff(t).astype(np.float32))


def oracle_mel_filterbank(sample_rate: int, n_fft: int, n_mel: int) -> torch.Tensor:
    """HTK mel scale, triangular, un-normalised, f_min 0 .. f_max sr/2: (n_fft/2+1, n_mel)."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (np.power(10.0, m / 2595.0) - 1.0)
    edges = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sample_rate / 2.0), n_mel + 2))      # n_mel+2 band edges in Hz
    bins = np.arange(n_fft // 2 + 1, dtype=np.float64) * (float(sample_rate // 2) / (n_fft // 2))
    fb = np.zeros((n_fft // 2 + 1, n_mel), dtype=np.float64)
    for m in range(n_mel):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        rise = (bins - lo) / (ce - lo)
        fall = (hi - bins) / (hi - ce)
        fb[:, m] = np.maximum(0.0, np.minimum(rise, fall))
    return torch.from_numpy(fb.astype(np.float32))


def oracle_hann(win_length: int) -> torch.Tensor:
    n = np.arange(win_length, dtype=np.float64)
    return torch.from_numpy((0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(np.float32))     # periodic


class Oracle:
    def __init__(self, spec, weights: Dict[str, torch.Tensor], nfe_step: int = 32, dtype: torch.dtype = torch.float32):
        """dtype = torch.float64 runs the SAME fp32 weights, tables and inputs through float64 arithmetic: the yardstick for what
        fp32 rounding alone costs at a given size (tests/test_fullsize_gpu.py derives its tolerance from it)."""
        self.spec = spec
        self.dt = dtype
        self.w = {k: v.to(torch.float32).to(dtype) for k, v in weights.items()}
        self.nfe_step = nfe_step
        self.t_grid, self.dt_grid = oracle_time_grid(nfe_step, spec.sway_coef)
        self.fb = oracle_mel_filterbank(spec.sample_rate, spec.n_fft, spec.n_mel).to(dtype)
        self.window = oracle_hann(spec.win_length).to(dtype)

    # ------------------------------------------------------------------ preprocess
    def mel(self, audio_i16: torch.Tensor) -> torch.Tensor:
        """int16 (S,) -> log-mel (S//hop+1, n_mel).  Frame count follows tts_engine.py:55."""
        s = self.spec
        x = audio_i16.to(self.dt) / 32768.0
        st = torch.stft(x, s.n_fft, hop_length=s.hop_length, win_length=s.win_length, window=self.window,
                        center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        mag = st.abs()                                  # (n_fft/2+1, frames)
        mel = self.fb.t() @ mag                         # (n_mel, frames)
        return mel.clamp(min=1e-5).log().t().contiguous()

    def text_pos_table(self, n: int) -> torch.Tensor:
        """[cos | sin] halves of pos * 10000^(-2i/d), float64 then fp32."""
        d = self.spec.text_dim
        ang = np.arange(n, dtype=np.float64)[:, None] * np.power(10000.0, -np.arange(0, d, 2, dtype=np.float64) / d)[None, :]
        return torch.from_numpy(np.concatenate([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)).to(self.dt)

    def grn(self, x: torch.Tensor, gamma, beta) -> torch.Tensor:
        gx = torch.norm(x, p=2, dim=0, keepdim=True)            # over the sequence, per channel
        nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
        return gamma * (x * nx) + beta + x

    def text_embed(self, text_ids: torch.Tensor, n: int, drop: bool) -> torch.Tensor:
        """ids (T,) -> (n, text_dim): +1 shift, truncate/pad with filler 0, embed, pos, ConvNeXtV2."""
        s, w = self.spec, self.w
        ids = text_ids.to(torch.long) + 1
        ids = ids[:n]
        ids = F.pad(ids, (0, n - ids.shape[0]), value=0)
        if drop:
            ids = torch.zeros_like(ids)
        x = w["text.embed.weight"][ids] + self.text_pos_table(n)
        for i in range(s.text_layers):
            p = f"text.blocks.{i}"
            r = x
            h = F.conv1d(x.t().unsqueeze(0), w[p + ".dwconv.weight"], w[p + ".dwconv.bias"],
                         padding=s.text_conv_k // 2, groups=s.text_dim).squeeze(0).t()
            h = F.layer_norm(h, (s.text_dim,), w[p + ".norm.weight"], w[p + ".norm.bias"], eps=1e-6)
            h = F.gelu(F.linear(h, w[p + ".pwconv1.weight"], w[p + ".pwconv1.bias"]))
            h = self.grn(h, w[p + ".grn.gamma"], w[p + ".grn.beta"])
            h = F.linear(h, w[p + ".pwconv2.weight"], w[p + ".pwconv2.bias"])
            x = r + h
        return x

    def rope_tables(self, n: int):
        """(cos_q, sin_q, cos_k, sin_k), each (n, head_dim): interleaved pairs (2i, 2i+1) share the angle
        pos * theta^(-2i/head_dim); the q tables carry the softmax scale head_dim^-0.5.  float64 then fp32."""
        s = self.spec
        hd = s.head_dim
        ang = np.arange(n, dtype=np.float64)[:, None] * np.power(float(s.rope_theta), -np.arange(0, hd, 2, dtype=np.float64) / hd)[None, :]
        ang = np.repeat(ang, 2, axis=1)
        scale = float(hd) ** -0.5
        f = lambda a: torch.from_numpy(a.astype(np.float32)).to(self.dt)
        return f(np.cos(ang) * scale), f(np.sin(ang) * scale), f(np.cos(ang)), f(np.sin(ang))

    def preprocess(self, audio_i16: torch.Tensor, text_ids: torch.Tensor, max_duration: int,
                   noise: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One utterance.  noise (N, n_mel) is an explicit input (the reference draws it inside
        preprocess.onnx under onnxruntime's seed, model.py:133 -- not reproducible outside ORT)."""
        s = self.spec
        n = int(max_duration)
        mel = self.mel(audio_i16)
        ref_len = mel.shape[0]
        assert ref_len <= n, "max_duration must cover the reference clip"
        cond = F.pad(mel, (0, 0, 0, n - ref_len))
        te = self.text_embed(text_ids, n, drop=False)
        te_drop = self.text_embed(text_ids, n, drop=True)
        cq, sq, ck, sk = self.rope_tables(n)
        return {
            "noise": noise.to(torch.float32).to(self.dt).clone(),
            "rope_cos_q": cq, "rope_sin_q": sq, "rope_cos_k": ck, "rope_sin_k": sk,
            "cat_mel_text": torch.cat([cond, te], dim=-1),
            "cat_mel_text_drop": torch.cat([torch.zeros_like(cond), te_drop], dim=-1),
            "ref_signal_len": ref_len,
        }

    # ------------------------------------------------------------------ transformer
    def time_embed(self, step: int) -> torch.Tensor:
        s, w = self.spec, self.w
        half = s.time_freq_dim // 2
        arg = 1000.0 * float(self.t_grid[step]) * np.exp(-np.arange(half, dtype=np.float64) * (math.log(10000.0) / (half - 1)))
        emb = torch.from_numpy(np.concatenate([np.sin(arg), np.cos(arg)]).astype(np.float32)).to(self.dt)      # [sin | cos], float64 then fp32
        h = F.silu(F.linear(emb, w["time.mlp1.weight"], w["time.mlp1.bias"]))
        return F.linear(h, w["time.mlp2.weight"], w["time.mlp2.bias"])

    @staticmethod
    def rope_apply(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        """x (n, heads, hd); cos/sin (n, hd); interleaved pairs."""
        x1 = x[..., 0::2]
        x2 = x[..., 1::2]
        rot = torch.stack((-x2, x1), dim=-1).flatten(-2)
        return x * cos.unsqueeze(1) + rot * sin.unsqueeze(1)

    def input_embed(self, x: torch.Tensor, cat: torch.Tensor) -> torch.Tensor:
        s, w = self.spec, self.w
        h = F.linear(torch.cat([x, cat], dim=-1), w["input.proj.weight"], w["input.proj.bias"])
        c = h.t().unsqueeze(0)
        pad = s.pos_conv_k // 2
        c = F.mish(F.conv1d(c, w["input.pos_conv1.weight"], w["input.pos_conv1.bias"], padding=pad,
                            groups=s.pos_conv_groups))
        c = F.mish(F.conv1d(c, w["input.pos_conv2.weight"], w["input.pos_conv2.bias"], padding=pad,
                            groups=s.pos_conv_groups))
        return c.squeeze(0).t() + h

    def attention(self, h: torch.Tensor, p: str, ropes) -> torch.Tensor:
        s, w = self.spec, self.w
        n = h.shape[0]
        qkv = F.linear(h, w[p + ".attn.qkv.weight"], w[p + ".attn.qkv.bias"])
        q, k, v = qkv.split(s.dim, dim=-1)
        q = self.rope_apply(q.reshape(n, s.heads, s.head_dim), ropes[0], ropes[1])   # scale folded in
        k = self.rope_apply(k.reshape(n, s.heads, s.head_dim), ropes[2], ropes[3])
        v = v.reshape(n, s.heads, s.head_dim)
        sc = torch.einsum("qhd,khd->hqk", q, k)
        pr = torch.softmax(sc, dim=-1)
        o = torch.einsum("hqk,khd->qhd", pr, v).reshape(n, s.dim)
        return F.linear(o, w[p + ".attn.out.weight"], w[p + ".attn.out.bias"])

    def dit_forward(self, x: torch.Tensor, cat: torch.Tensor, ropes, step: int,
                    n_blocks: Optional[int] = None) -> torch.Tensor:
        """x (N, n_mel), cat (N, cond_dim) -> predicted flow (N, n_mel)."""
        s, w = self.spec, self.w
        d = s.dim
        temb = F.silu(self.time_embed(step))
        h = self.input_embed(x, cat)
        for i in range(s.depth if n_blocks is None else n_blocks):
            p = f"blocks.{i}"
            mod = F.linear(temb, w[p + ".adaln.weight"], w[p + ".adaln.bias"])
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6)
            a = F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_a) + sh_a
            h = h + g_a * self.attention(a, p, ropes)
            m = F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_m) + sh_m
            m = F.gelu(F.linear(m, w[p + ".ff1.weight"], w[p + ".ff1.bias"]), approximate="tanh")
            h = h + g_m * F.linear(m, w[p + ".ff2.weight"], w[p + ".ff2.bias"])
        mod = F.linear(temb, w["final.adaln.weight"], w["final.adaln.bias"])
        sc_f, sh_f = mod.chunk(2)
        h = F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_f) + sh_f
        return F.linear(h, w["final.proj.weight"], w["final.proj.bias"])

    def transformer_step(self, x: torch.Tensor, pre: Dict[str, torch.Tensor], step: int) -> torch.Tensor:
        ropes = (pre["rope_cos_q"], pre["rope_sin_q"], pre["rope_cos_k"], pre["rope_sin_k"])
        pc = self.dit_forward(x, pre["cat_mel_text"], ropes, step)
        pu = self.dit_forward(x, pre["cat_mel_text_drop"], ropes, step)
        pred = pc + (pc - pu) * self.spec.cfg_strength
        return x + pred * float(self.dt_grid[step])

    # ------------------------------------------------------------------ decode
    def vocoder(self, mel: torch.Tensor) -> torch.Tensor:
        """mel (T, n_mel) -> waveform float (T*hop,) in (-1,1)."""
        s, w = self.spec, self.w
        x = mel.t().unsqueeze(0)
        x = F.conv1d(x, w["voc.pre.weight"], w["voc.pre.bias"], padding=s.voc_pre_k // 2)
        for st, (r, k) in enumerate(zip(s.voc_up_rates, s.voc_up_kernels)):
            x = F.leaky_relu(x, s.voc_lrelu)
            x = F.conv_transpose1d(x, w[f"voc.up.{st}.weight"], w[f"voc.up.{st}.bias"], stride=r,
                                   padding=(k - r) // 2)
            acc = None
            for a, rk in enumerate(s.voc_res_kernels):
                y = x
                for b, dil in enumerate(s.voc_res_dilations):
                    q = f"voc.res.{st}.{a}.{b}"
                    t = F.leaky_relu(y, s.voc_lrelu)
                    t = F.conv1d(t, w[q + ".conv1.weight"], w[q + ".conv1.bias"], dilation=dil,
                                 padding=dil * (rk - 1) // 2)
                    t = F.leaky_relu(t, s.voc_lrelu)
                    t = F.conv1d(t, w[q + ".conv2.weight"], w[q + ".conv2.bias"], padding=(rk - 1) // 2)
                    y = t + y
                acc = y if acc is None else acc + y
            x = acc / len(s.voc_res_kernels)
        x = F.leaky_relu(x, 0.01)
        x = F.conv1d(x, w["voc.post.weight"], w["voc.post.bias"], padding=s.voc_post_k // 2)
        return torch.tanh(x).reshape(-1)

    @staticmethod
    def to_pcm(wave: torch.Tensor) -> torch.Tensor:
        return torch.clamp(wave * 32767.0, -32768.0, 32767.0).to(torch.int16)   # truncation toward 0

    def decode(self, x: torch.Tensor, ref_signal_len: int) -> torch.Tensor:
        mel = x[int(ref_signal_len):]
        return self.to_pcm(self.vocoder(mel))

    # ------------------------------------------------------------------ whole utterance
    def synthesize(self, audio_i16, text_ids, max_duration, noise, n_steps: Optional[int] = None):
        pre = self.preprocess(audio_i16, text_ids, max_duration, noise)
        x = pre["noise"]
        for st in range(self.nfe_step - 1 if n_steps is None else n_steps):
            x = self.transformer_step(x, pre, st)
        return x, self.decode(x, pre["ref_signal_len"])


class OracleSession:
    """A session object with onnxruntime's ``run(output_names, feed)`` shape, backed by the oracle.
    Used by the CPU plumbing tests (BASELINE config 1) to stand where the reference's
    onnxruntime.InferenceSession stands (core/model.py:98-106).  Batch 1, like the reference."""

    def __init__(self, oracle: Oracle, kind: str, seed: int = 9527):
        self.oracle, self.kind = oracle, kind
        self.gen = torch.Generator().manual_seed(seed)
        self._pre = None

    def input_names(self) -> List[str]:
        return {"preprocess": ["audio", "text_ids", "max_duration"],
                "transformer": ["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k",
                                "cat_mel_text", "cat_mel_text_drop", "time_step"],
                "decode": ["denoised", "ref_signal_len"]}[self.kind]

    def output_names(self) -> List[str]:
        return {"preprocess": ["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k",
                               "cat_mel_text", "cat_mel_text_drop", "ref_signal_len"],
                "transformer": ["denoised", "time_step_out"],
                "decode": ["output_audio"]}[self.kind]

    def get_inputs(self):
        return [type("Io", (), {"name": n})() for n in self.input_names()]

    def get_outputs(self):
        return [type("Io", (), {"name": n})() for n in self.output_names()]

    def run(self, output_names, feed):
        import numpy as np
        vals = [feed[n] for n in self.input_names()]
        o = self.oracle
        if self.kind == "preprocess":
            audio, text_ids, max_dur = vals
            n = int(np.asarray(max_dur).reshape(-1)[0])
            noise = torch.randn((n, o.spec.n_mel), generator=self.gen, dtype=torch.float32)
            pre = o.preprocess(torch.from_numpy(np.asarray(audio).reshape(-1)),
                               torch.from_numpy(np.asarray(text_ids).reshape(-1)), n, noise)
            outs = []
            for name in self.output_names():
                v = pre[name]
                if name == "ref_signal_len":
                    outs.append(np.array([v], dtype=np.int64))
                elif name.startswith("rope"):
                    outs.append(v.numpy()[None])
                else:
                    outs.append(v.numpy()[None])
            return outs
        if self.kind == "transformer":
            x, cq, sq, ck, sk, cat, catd, ts = [np.asarray(v) for v in vals]
            step = int(ts.reshape(-1)[0])
            pre = {"rope_cos_q": torch.from_numpy(cq[0]), "rope_sin_q": torch.from_numpy(sq[0]),
                   "rope_cos_k": torch.from_numpy(ck[0]), "rope_sin_k": torch.from_numpy(sk[0]),
                   "cat_mel_text": torch.from_numpy(cat[0]), "cat_mel_text_drop": torch.from_numpy(catd[0])}
            y = o.transformer_step(torch.from_numpy(x[0]), pre, step)
            return [y.numpy()[None], np.array([step + 1], dtype=np.int32)]
        x, ref_len = vals
        pcm = o.decode(torch.from_numpy(np.asarray(x)[0]), int(np.asarray(ref_len).reshape(-1)[0]))
        return [pcm.numpy().reshape(1, 1, -1)]

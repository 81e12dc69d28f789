"""The oracle's constant tables vs the product's (CPU, no GPU).

oracle/vv_oracle.py builds its mel filterbank, Hann window, sway time grid, sinusoidal time table, rope and text
position tables from its own float64 closed forms (numpy) and imports nothing from vietvoice_tts_amd; the product
builds its tables in model_spec.py / pack.py (torch).  A wrong formula on either side shows up here instead of
cancelling out inside every HIP-vs-oracle parity test.  Tolerance: one fp32 rounding (tables are defined in float64
and rounded once), i.e. 2 ulp of the table's magnitude.
"""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import vv_oracle
from vietvoice_tts_amd import pack
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights, mel_filterbank, time_grid

ULP = float(np.finfo(np.float32).eps)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(a: torch.Tensor, b: torch.Tensor, scale: float = 1.0, ulps: float = 2.0):
    assert a.shape == b.shape and a.dtype == b.dtype == torch.float32
    err = float((a.double() - b.double()).abs().max())
    assert err <= ulps * ULP * scale, err


def test_oracle_imports_nothing_from_the_product():
    tree = ast.parse(open(os.path.join(ROOT, "oracle", "vv_oracle.py")).read())
    mods = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Import):
            mods += [a.name for a in n.names]
        elif isinstance(n, ast.ImportFrom):
            mods.append(n.module or "")
    assert not [m for m in mods if m.startswith("vietvoice")], mods


@pytest.mark.parametrize("spec", [ModelSpec.full(), ModelSpec.tiny()], ids=["full", "tiny"])
def test_mel_filterbank_and_window(spec):
    fb_o = vv_oracle.oracle_mel_filterbank(spec.sample_rate, spec.n_fft, spec.n_mel)
    fb_p = mel_filterbank(spec)
    _close(fb_o, fb_p, scale=1.0, ulps=4.0)
    # known structure: triangles, every filter non-empty, peak <= 1, band centres increasing
    assert float(fb_o.max()) <= 1.0 and bool((fb_o.sum(0) > 0).all())
    assert bool((fb_o.argmax(0)[1:] >= fb_o.argmax(0)[:-1]).all())
    _close(vv_oracle.oracle_hann(spec.win_length), torch.hann_window(spec.win_length, periodic=True, dtype=torch.float32))


@pytest.mark.parametrize("nfe", [2, 8, 32])
def test_time_grid(nfe):
    spec = ModelSpec.full()
    t_o, dt_o = vv_oracle.oracle_time_grid(nfe, spec.sway_coef)
    t_p, dt_p = time_grid(nfe, spec.sway_coef)
    _close(t_o, t_p)
    _close(dt_o, dt_p)
    assert t_o.numel() == nfe - 1 and float(t_o[0]) == 0.0          # nfe_step-1 Euler steps (reference tts_engine.py:157-159)
    assert abs(float(t_o[-1] + dt_o[-1]) - 1.0) < 1e-6 and bool((dt_o > 0).all())
    # sway sampling with coefficient -1: t = 1 - cos(pi/2 u)
    u = np.arange(nfe) / (nfe - 1)
    assert np.abs(t_o.numpy() - (1 - np.cos(np.pi / 2 * u))[:-1]).max() < 1e-6


@pytest.mark.parametrize("spec", [ModelSpec.full(), ModelSpec.small()], ids=["full", "small"])
def test_rope_text_and_time_tables(spec):
    w = {k: v for k, v in make_synthetic_weights(spec).items() if k.startswith("time.")}
    orc = vv_oracle.Oracle(spec, w, nfe_step=32)
    n = 1900
    for o, p in zip(orc.rope_tables(n), pack.rope_tables(spec, n)):
        _close(o, p)
    cq, sq, ck, sk = orc.rope_tables(n)
    assert bool(torch.equal(ck[:, 0::2], ck[:, 1::2]))                 # interleaved pairs share one angle
    assert abs(float(cq[0, 0]) - spec.head_dim ** -0.5) < 1e-7 and float(sk[0].abs().max()) == 0.0
    assert abs(float(ck[1, 0]) - np.cos(1.0)) < 1e-7                   # pair 0 rotates by exactly pos radians
    _close(orc.text_pos_table(n), pack.text_pos_table(spec, n))
    sin_tab = pack.time_sinus_table(spec, orc.t_grid)                  # product table on the oracle's grid
    half = spec.time_freq_dim // 2
    for step in (0, 7, 30):
        arg = 1000.0 * float(orc.t_grid[step]) * np.exp(-np.arange(half) * (np.log(10000.0) / (half - 1)))
        ref = torch.from_numpy(np.concatenate([np.sin(arg), np.cos(arg)]).astype(np.float32))
        _close(sin_tab[step], ref)
        # and the oracle's whole time MLP consumes exactly that table
        h = torch.nn.functional.silu(torch.nn.functional.linear(ref, w["time.mlp1.weight"], w["time.mlp1.bias"]))
        assert torch.allclose(orc.time_embed(step), torch.nn.functional.linear(h, w["time.mlp2.weight"], w["time.mlp2.bias"]), atol=1e-6)


def test_bf16_yardstick_is_the_float64_oracle_plus_roundings(monkeypatch):
    """oracle/vv_oracle_bf16.py restates the model's data flow with bf16 roundings at the bf16 model's storage points.  With the
    rounding function replaced by the identity it must BE the float64 oracle (same graph, same order of additions); with it, the
    result moves by a bf16-class amount and not more."""
    from oracle import vv_oracle_bf16 as yb
    from oracle.vv_oracle import Oracle
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    spec = ModelSpec.tiny()
    w = make_synthetic_weights(spec, 9527)
    g = torch.Generator().manual_seed(3)
    S, T, gen = 256 * 14, 20, 10
    audio = (torch.randn(S, generator=g) * 3000).to(torch.int16)
    ids = torch.randint(0, spec.vocab_size, (T,), generator=g, dtype=torch.int32)
    N = S // spec.hop_length + 1 + gen
    noise = torch.randn(N, spec.n_mel, generator=g)
    with torch.no_grad():
        x64, _ = Oracle(spec, w, nfe_step=6, dtype=torch.float64).synthesize(audio, ids, N, noise)
        xb, _ = yb.Bf16Oracle(spec, w, nfe_step=6).synthesize(audio, ids, N, noise)
        monkeypatch.setattr(yb, "rb", lambda t: t)
        xi, _ = yb.Bf16Oracle(spec, w, nfe_step=6).synthesize(audio, ids, N, noise)
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    assert rel(xi, x64) < 1e-12, rel(xi, x64)
    assert 1e-5 < rel(xb, x64) < 3e-2, rel(xb, x64)

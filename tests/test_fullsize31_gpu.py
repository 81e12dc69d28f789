"""-m gpu: the PRODUCTION run of the hot path at BASELINE's full model size -- the reference's 31 transformer evaluations over
the 32-point sway grid (/root/reference/vietvoicetts/core/tts_engine.py:157-159: range(0, nfe_step - 1, fuse_nfe), nfe_step = 32,
fuse_nfe = 1) followed by one decode (:176-187) -- against the committed oracle fixture tests/golden/fullsize_golden.{npz,json}
(generator: tests/golden/make_fullsize_golden.py, run in the build container; the float64 oracle needs 8 minutes per utterance,
too slow for this box's test run).

Inputs are NOT in the fixture: they are bench.py's seeded item 0 and the seeded synthetic weights, regenerated here and checked
against the digests the generator stored -- a generator that changed silently fails the digest check, not the tolerance.

Tolerances follow tests/test_fullsize_gpu.py: the fixture carries how far the plain-torch fp32 oracle is from the float64 one at
every kept step (after 31 steps: max |err| 4.7e-4, rmse / rms 2.3e-5 on a state of range 12, PCM +-1 LSB); the HIP fp32 path is
held to a small multiple of THAT (it cannot be held to SURVEY's 1e-4: no fp32 implementation meets it), the bf16 path to 2 x the
figures measured when the test was written.  PARITY UNPINNED against the real reference graphs (oracle/vv_oracle.py header).
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"


def _digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


@pytest.fixture(scope="module")
def gold():
    import bench
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    with open(os.path.join(GOLD, "fullsize_golden.json")) as fh:
        meta = json.load(fh)
    arr = np.load(os.path.join(GOLD, "fullsize_golden.npz"))
    spec = ModelSpec.full()
    w = make_synthetic_weights(spec, bench.SEED)
    d, N = bench.make_inputs(spec, 32, 0, "cpu")
    assert N == meta["N"] == 1600 and meta["nfe_step"] == 32 and meta["euler_steps"] == 31 and meta["truth"] == "f64"
    # the seeded generators reproduce the generator's bytes on this machine
    assert _digest(d["audio"][0]) == meta["inputs"]["audio0"] and _digest(d["ids"][0]) == meta["inputs"]["ids0"]
    assert _digest(d["noise"][0]) == meta["inputs"]["noise0"] and len(w) == meta["inputs"]["n_weights"]
    for k, h in meta["inputs"]["weights"].items():
        assert _digest(w[k]) == h, k
    return dict(meta=meta, arr=arr, spec=spec, w=w, d=d, N=N)


def _run31(eng, d, N, keep):
    """All 31 Euler steps through vv_transformer_steps_h in the segments between the kept steps, then the decode."""
    import bench
    one = {k: v[0:1].contiguous().to(DEV) for k, v in d.items() if torch.is_tensor(v)}
    pre = eng.preprocess(one["audio"], one["audio_len"], one["ids"], one["text_len"], one["seq_len"], N, seq_len_host=[N])
    x = one["noise"].clone()
    states, st = {}, 0
    for k in keep:
        eng.transformer_steps(x, pre, st, k - st)
        st = k
        states[k] = x[0].cpu().double()
    assert st == eng.n_steps == 31
    pcm, pcm_len, wave = eng.decode(x, pre, bench.GEN_FRAMES, want_wave=True)
    torch.cuda.synchronize()
    return pre, states, pcm[0].cpu(), int(pcm_len[0]), wave[0].cpu().double()


def test_fp32_production_run_matches_the_oracle_fixture(gold):
    """configs[1] with the production step count: B = 1, N = 1600, fp32 acoustic + fp32 vocoder, 31 steps."""
    from vietvoice_tts_amd.runtime import HipSynth
    g, meta = gold, gold["meta"]
    keep = meta["keep_steps"]
    eng = HipSynth(g["spec"], g["w"], acoustic_dtype="fp32", nfe_step=32)
    pre, states, pcm, pcm_len, wave = _run31(eng, g["d"], g["N"], keep)
    eng.close()
    assert int(pre["ref_signal_len"][0]) == meta["ref_signal_len"] == 563
    checks = []
    for k in keep:
        ref = torch.from_numpy(g["arr"][f"x{k}"]).double()
        o32 = meta["torch_fp32_vs_f64"]["state"][str(k)]
        err = (states[k] - ref).abs()
        mx, rm = float(err.max()), float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"\n[full fp32, 31 steps] after step {k:2d} (state range {o32['state_max']:.2f}): HIP max|err| {mx:.2e} rmse/rms {rm:.2e} | "
              f"torch-fp32 oracle max|err| {o32['max_err']:.2e} rmse/rms {o32['rmse_over_rms']:.2e}")
        # the fixture's states are the float64 run rounded once to fp32 (6e-8 relative): floors keep the first, tiny step honest
        checks.append((f"state max err after step {k}", mx, 3.0 * o32["max_err"] + 2e-6))
        checks.append((f"state rmse/rms after step {k}", rm, 2.0 * o32["rmse_over_rms"] + 2e-7))
    n = g["arr"]["pcm"].size
    assert pcm_len == n == 1037 * g["spec"].hop_length
    ref_w = torch.from_numpy(g["arr"]["wave"]).double()
    tw = meta["torch_fp32_vs_f64"]
    e_w = float((wave[:n] - ref_w).abs().max())
    dp = (pcm[:n].int() - torch.from_numpy(g["arr"]["pcm"]).int()).abs()
    print(f"[full fp32, 31 steps] waveform max abs err {e_w:.2e} (torch-fp32 oracle {tw['wave_max_err']:.2e}, peak {tw['wave_peak']:.3f}); "
          f"PCM max diff {int(dp.max())} LSB, {int((dp > 0).sum())} of {n} samples differ (torch-fp32 oracle: max {tw['pcm_max_lsb']} LSB)")
    checks += [("waveform", e_w, 3.0 * tw["wave_max_err"]), ("pcm lsb", int(dp.max()), tw["pcm_max_lsb"] + 1),
               ("pcm share beyond 1 LSB", float((dp > 1).float().mean()), 1e-4)]
    bad = [c for c in checks if not c[1] <= c[2]]
    assert not bad, bad


def test_bf16_production_run_close_to_the_oracle_fixture(gold):
    """configs[2]'s arithmetic (bf16 acoustic + fp32-fidelity vocoder) at B = 1 with the production step count."""
    from vietvoice_tts_amd.runtime import HipSynth
    g, meta = gold, gold["meta"]
    keep = meta["keep_steps"]
    eng = HipSynth(g["spec"], g["w"], acoustic_dtype="bf16", nfe_step=32)
    _pre, states, pcm, pcm_len, wave = _run31(eng, g["d"], g["N"], keep)
    eng.close()
    # bounds = 2 x the figures measured when the test was written (round 3; BF16_MEASURED below), so a regression that doubles
    # the bf16 error fails
    got = {}
    for k in keep:
        ref = torch.from_numpy(g["arr"][f"x{k}"]).double()
        err = (states[k] - ref).abs()
        got[k] = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"\n[full bf16, 31 steps] after step {k:2d}: state rmse/rms {got[k]:.3e}, max abs err {float(err.max()):.3e}")
    n = g["arr"]["pcm"].size
    ref_w = torch.from_numpy(g["arr"]["wave"]).double()
    wr = float((wave[:n] - ref_w).pow(2).mean().sqrt() / ref_w.pow(2).mean().sqrt())
    print(f"[full bf16, 31 steps] waveform rmse/rms {wr:.3e}")
    assert pcm_len == n and bool(torch.isfinite(wave).all()) and int(pcm[:n].abs().max()) > 0
    for k in keep:
        assert got[k] <= 2.0 * BF16_MEASURED[k], (k, got[k])
    assert wr <= 2.0 * BF16_MEASURED["wave"], wr


# state rmse/rms after steps 1, 8, 16, 24, 31 and waveform rmse/rms of the bf16 run above, as first measured (round 3)
BF16_MEASURED = {1: 1.83e-5, 8: 9.02e-4, 16: 2.91e-3, 24: 4.49e-3, 31: 5.17e-3, "wave": 5.06e-3}

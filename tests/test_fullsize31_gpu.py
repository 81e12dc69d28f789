"""-m gpu: the PRODUCTION run of the hot path at BASELINE's full model size -- the reference's 31 transformer evaluations over
the 32-point sway grid (/root/reference/vietvoicetts/core/tts_engine.py:157-159: range(0, nfe_step - 1, fuse_nfe), nfe_step = 32,
fuse_nfe = 1) followed by one decode (:176-187) -- against the committed oracle fixture tests/golden/fullsize_golden.{npz,json}
(generator: tests/golden/make_fullsize_golden.py, run in the build container; the float64 oracle needs 8 minutes per utterance,
too slow for this box's test run).

Inputs are NOT in the fixture: they are bench.py's seeded item 0 and the seeded synthetic weights, regenerated here and checked
against the digests the generator stored -- a generator that changed silently fails the digest check, not the tolerance.

Tolerances follow tests/test_fullsize_gpu.py: the fixture carries how far the plain-torch fp32 oracle is from the float64 one at
every kept step (after 31 steps: max |err| 4.7e-4, rmse / rms 2.3e-5 on a state of range 12, PCM +-1 LSB); the HIP fp32 path is
held to a small multiple of THAT (it cannot be held to SURVEY's 1e-4: no fp32 implementation meets it).  The bf16 path is held to an
INDEPENDENT yardstick since round 5 (tests/golden/fullsize_bf16_yardstick.*): the float64 oracle with bf16 roundings at the bf16
model's storage points (oracle/vv_oracle_bf16.py), all 31 steps of item 0 -- what the bf16 FORMAT costs, measured without any
kernel of the product (rounds 3-4 used "2 x what the same kernels produced").  PARITY UNPINNED against the real reference graphs
(oracle/vv_oracle.py header).
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


@pytest.fixture(scope="module")
def yard(gold):
    with open(os.path.join(GOLD, "fullsize_bf16_yardstick.json")) as fh:
        ym = json.load(fh)
    assert ym["inputs"] == gold["meta"]["inputs"] and ym["nfe_step"] == 32 and ym["item"] == 0 and ym["steps"] == 31
    state = {int(k): v["rmse_over_rms"] for k, v in ym["yardstick_vs_f64"].items()}
    return dict(meta=ym, arr=np.load(os.path.join(GOLD, "fullsize_bf16_yardstick.npz")), state=state,
                wave=ym["yardstick_wave_vs_f64"]["rmse_over_rms"], pcm=ym["yardstick_wave_vs_f64"]["pcm_rmse_over_rms"])


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


def test_bf16_production_run_is_the_price_of_the_format(gold, yard):
    """configs[2]'s arithmetic (bf16 acoustic + fp32-fidelity vocoder) at B = 1 with the production step count, against the float64
    fixture -- and the error it may have is the independent yardstick's (the float64 oracle with bf16 storage roundings): within
    +-10 % of it at every kept step and at the waveform (measured round 5: 1.834e-5 / 9.016e-4 / 2.91e-3 / 4.49e-3 / 5.17e-3 and 5.06e-3
    against the yardstick's 1.832e-5 / 9.046e-4 / 2.919e-3 / 4.507e-3 / 5.191e-3 and 5.080e-3: the same to three digits).  Two-sided:
    more than the format's price is a defect, far less means the run is not the bf16 model."""
    from vietvoice_tts_amd.runtime import HipSynth
    g, meta = gold, gold["meta"]
    keep = meta["keep_steps"]
    eng = HipSynth(g["spec"], g["w"], acoustic_dtype="bf16", nfe_step=32)
    _pre, states, pcm, pcm_len, wave = _run31(eng, g["d"], g["N"], keep)
    eng.close()
    bad = []
    for k in keep:
        ref = torch.from_numpy(g["arr"][f"x{k}"]).double()
        err = (states[k] - ref).abs()
        got = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"\n[full bf16, 31 steps] after step {k:2d}: state rmse/rms {got:.3e} (yardstick {yard['state'][k]:.3e}), max abs err {float(err.max()):.3e}")
        if not 0.9 * yard["state"][k] - 2e-7 <= got <= 1.1 * yard["state"][k] + 2e-7:
            bad.append((k, got, yard["state"][k]))
    n = g["arr"]["pcm"].size
    ref_w = torch.from_numpy(g["arr"]["wave"]).double()
    wr = float((wave[:n] - ref_w).pow(2).mean().sqrt() / ref_w.pow(2).mean().sqrt())
    print(f"[full bf16, 31 steps] waveform rmse/rms {wr:.3e} (yardstick {yard['wave']:.3e})")
    assert pcm_len == n and bool(torch.isfinite(wave).all()) and int(pcm[:n].abs().max()) > 0
    if not 0.9 * yard["wave"] <= wr <= 1.1 * yard["wave"]:
        bad.append(("wave", wr, yard["wave"]))
    assert not bad, bad


def test_bf16_run_stays_near_the_yardstick_trajectory(gold, yard):
    """VERDICT r4 #8, the second half: besides costing what the format costs (the test above), the HIP bf16 run must stay NEAR the
    yardstick's own trajectory: after steps 1 / 4 / 8 / 12 its distance to the yardstick state is at most 1.2 x the format's price
    (measured 0.69 - 0.94 x: two realisations of one rounding noise decorrelate as a perturbation flips roundings downstream and
    approach sqrt(2) x only for a path with a DIFFERENT error mechanism; a systematic deviation shows here first), and its error
    against float64 is the yardstick's within +-10 % at the intermediate steps the 31-step test does not keep."""
    from vietvoice_tts_amd.runtime import HipSynth
    g, ym, ya = gold, yard["meta"], yard["arr"]
    keep = [k for k in ym["array_steps"] if k <= 12]
    one = {k: v[0:1].contiguous().to(DEV) for k, v in g["d"].items() if torch.is_tensor(v)}
    eng = HipSynth(g["spec"], g["w"], acoustic_dtype="bf16", nfe_step=32)
    pre = eng.preprocess(one["audio"], one["audio_len"], one["ids"], one["text_len"], one["seq_len"], g["N"], seq_len_host=[g["N"]])
    x, st, bad = one["noise"].clone(), 0, []
    rel = lambda got, ref: float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    for k in keep:
        eng.transformer_steps(x, pre, st, k - st)
        st = k
        got = x[0].cpu().double()
        src = ym["f64_states"][str(k)]                       # the float64 state: in this fixture, or the one fullsize_golden.npz already holds
        f64 = torch.from_numpy(g["arr"][src.split(":")[1]] if ":" in src else ya[src]).double()
        yd = torch.from_numpy(ya[f"yard_x{k}"]).double()
        e_hip, e_yard, d_hy = rel(got, f64), yard["state"][k], rel(got, yd)
        print(f"\n[full bf16 vs yardstick] after step {k:2d}: HIP - f64 {e_hip:.3e} | yardstick - f64 {e_yard:.3e} | HIP - yardstick {d_hy:.3e}")
        if not 0.9 * e_yard - 2e-7 <= e_hip <= 1.1 * e_yard + 2e-7:
            bad.append(("error vs float64 is not the format's price", k, e_hip, e_yard))
        if not d_hy <= 1.2 * e_yard + 2e-7:
            bad.append(("too far from the yardstick", k, d_hy, e_yard))
    eng.close()
    assert not bad, bad


def test_bf16_headline_batch_item0_production_run(gold, yard):
    """configs[2] itself: the headline batch (B = 32, bench inputs, bf16 acoustic + fp32-fidelity vocoder) through
    `synthesize_batch` -- the call bench.py times -- with the production step count; item 0 of the batch against the oracle fixture
    (same bounds as the single-utterance run: rows are packed and every kernel is row- or sequence-local), the whole batch finite,
    full length and not one utterance repeated; then item 31 -- the last rows of every launch -- against its own fixture, with and
    without the split-K tail option.  Bounds: the independent bf16 yardstick's figures for item 0 (tests/golden/
    fullsize_bf16_yardstick.*): item 0 within +-10 %; another item (31: other text, other clip, other noise) at most 1.15 x (its own
    error measured 3 % above item 0's)."""
    import bench
    from vietvoice_tts_amd.runtime import HipSynth
    g = gold
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in g["d"].items()}
    eng = HipSynth(g["spec"], g["w"], acoustic_dtype="bf16", nfe_step=32)
    x, pcm, pcm_len, _pre = eng.synthesize_batch(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], g["N"], d["noise"], bench.GEN_FRAMES,
                                                 seq_len_host=d["seq_len_host"])
    torch.cuda.synchronize()
    eng.close()
    n = g["arr"]["pcm"].size
    assert bool((pcm_len == n).all()) and bool(torch.isfinite(x).all())
    ref = torch.from_numpy(g["arr"]["x31"]).double()
    got = x[0].cpu().double()
    rm = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    ref_pcm = torch.from_numpy(g["arr"]["pcm"]).double()
    pr = float((pcm[0, :n].cpu().double() - ref_pcm).pow(2).mean().sqrt() / ref_pcm.pow(2).mean().sqrt())
    print(f"\n[full bf16 B=32, 31 steps] item 0 vs the float64 fixture: state rmse/rms {rm:.3e}, PCM rmse/rms {pr:.3e}")
    assert 0.9 * yard["state"][31] <= rm <= 1.1 * yard["state"][31] and 0.9 * yard["pcm"] <= pr <= 1.1 * yard["pcm"], (rm, pr)
    assert float((x[1] - x[0]).abs().max()) > 1e-2 and float(pcm.float().abs().amax(dim=1).min()) > 0
    # ---- item 31, the LAST item of the batch: its unconditional rows are the last rows of every launch (rows 100,800 .. 102,399 of
    # 102,400: the last row panels of the persistent GEMM, the panels a split-K tail would take), against ITS OWN float64 fixture
    # (tests/golden/fullsize_golden_item31.*, generator `make_fullsize_golden.py --item 31`): every item is an independent B = 1 call
    # in the reference (/root/reference/vietvoicetts/core/tts_engine.py:47,121), so the last item is held to the first one's bounds.
    with open(os.path.join(GOLD, "fullsize_golden_item31.json")) as fh:
        m31 = json.load(fh)
    a31 = np.load(os.path.join(GOLD, "fullsize_golden_item31.npz"))
    assert m31["item"] == 31 and m31["truth"] == "f64" and m31["nfe_step"] == 32
    assert _digest(g["d"]["audio"][31]) == m31["inputs"]["audio0"] and _digest(g["d"]["ids"][31]) == m31["inputs"]["ids0"]
    assert _digest(g["d"]["noise"][31]) == m31["inputs"]["noise0"]
    ref31 = torch.from_numpy(a31["x31"]).double()
    rm31 = float((x[31].cpu().double() - ref31).pow(2).mean().sqrt() / ref31.pow(2).mean().sqrt())
    rp31 = torch.from_numpy(a31["pcm"]).double()
    pr31 = float((pcm[31, :n].cpu().double() - rp31).pow(2).mean().sqrt() / rp31.pow(2).mean().sqrt())
    print(f"[full bf16 B=32, 31 steps] item 31 (last rows of every launch) vs its float64 fixture: state rmse/rms {rm31:.3e}, PCM rmse/rms {pr31:.3e}")
    assert rm31 <= 1.15 * yard["state"][31] and pr31 <= 1.15 * yard["pcm"], (rm31, pr31)
    # ---- the same item with the split-K tail OPTION on (FF2): its unconditional rows then really are tail rows (K parts summed by the
    # LayerNorm) -- the tail's arithmetic against the oracle, not only against the plain launch
    eng = HipSynth(g["spec"], g["w"], acoustic_dtype="bf16", nfe_step=32)
    eng.set_option("split_k_tail", 2)
    xt, pcmt, _lt, _p = eng.synthesize_batch(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], g["N"], d["noise"], bench.GEN_FRAMES,
                                             seq_len_host=d["seq_len_host"])
    torch.cuda.synchronize()
    eng.close()
    rmt = float((xt[31].cpu().double() - ref31).pow(2).mean().sqrt() / ref31.pow(2).mean().sqrt())
    prt = float((pcmt[31, :n].cpu().double() - rp31).pow(2).mean().sqrt() / rp31.pow(2).mean().sqrt())
    print(f"[full bf16 B=32, 31 steps] item 31 with split_k_tail = 2: state rmse/rms {rmt:.3e}, PCM rmse/rms {prt:.3e}; item 0 unchanged: {bool(torch.equal(xt[0], x[0]))}")
    assert rmt <= 1.15 * yard["state"][31] and prt <= 1.15 * yard["pcm"], (rmt, prt)
    assert torch.equal(xt[0], x[0]) and not torch.equal(xt[31], x[31])       # item 0 never enters a tail; item 31 does


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_ragged_batch_production_run(dtype, yard):
    """BASELINE configs[3] semantics with the production step count at FULL size: three utterances of different reference-clip, text
    and frame lengths as ONE packed ragged batch (per-item masks in attention / pos-conv / text conv, packed GEMM rows, bucketed
    vocoder), all 31 steps; EVERY item (0: 1600 frames at row_start 0, 1: 851, 2: 411) against the float64 oracle run on that item
    ALONE (tests/golden/fullsize_ragged_golden.*, generator `make_fullsize_golden.py --ragged [--items 0]`).  fp32: bounds from the
    torch-fp32 yardstick in the fixture; bf16 (configs[3]'s arithmetic): at most 1.15 x what the bf16 FORMAT costs on the headline item
    (the independent yardstick of tests/golden/fullsize_bf16_yardstick.*: 5.19e-3 state / 5.08e-3 PCM; these items measured
    4.93 - 5.09e-3 / 5.08 - 5.10e-3: a packed ragged batch costs no accuracy)."""
    import importlib.util
    import bench  # noqa: F401  (the generator module imports it)
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    spec_ = importlib.util.spec_from_file_location("vv_make_fullsize_golden", os.path.join(GOLD, "make_fullsize_golden.py"))
    gen = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(gen)
    with open(os.path.join(GOLD, "fullsize_ragged_golden.json")) as fh:
        meta = json.load(fh)
    arr = np.load(os.path.join(GOLD, "fullsize_ragged_golden.npz"))
    spec = ModelSpec.full()
    w = make_synthetic_weights(spec, bench.SEED)
    audio, ids, seq, noise = gen.ragged_inputs(spec)
    assert seq == meta["seq"] and _digest(audio) == meta["inputs"]["audio"] and _digest(ids) == meta["inputs"]["ids"] and _digest(noise) == meta["inputs"]["noise"]
    la, lt, gf = meta["la"], meta["lt"], meta["gf"]
    N = max(seq)
    eng = HipSynth(spec, w, acoustic_dtype=dtype, nfe_step=32)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    pre = eng.preprocess(audio.to(DEV), i32(la), ids.to(DEV), i32(lt), i32(seq), N, seq_len_host=seq)
    x = noise.to(DEV).clone()
    eng.transformer_steps(x, pre, 0, 31)
    pcm, pcm_len = eng.decode_bucketed(x, pre, gf, pad_frac=0.10, min_units=1)
    torch.cuda.synchronize()
    eng.close()
    checks = []
    assert sorted(meta["items"]) == ["0", "1", "2"]
    for b in (0, 1, 2):
        it = meta["items"][str(b)]
        y = it["torch_fp32_vs_f64"]
        ref = torch.from_numpy(arr[f"x31_{b}"]).double()
        got = x[b, : seq[b]].cpu().double()
        err = (got - ref).abs()
        mx, rm = float(err.max()), float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        n = arr[f"pcm_{b}"].size
        dp = (pcm[b, :n].cpu().int() - torch.from_numpy(arr[f"pcm_{b}"]).int()).abs()
        rp = torch.from_numpy(arr[f"pcm_{b}"]).double()
        pr = float((pcm[b, :n].cpu().double() - rp).pow(2).mean().sqrt() / rp.pow(2).mean().sqrt())
        print(f"\n[full {dtype} ragged, 31 steps] item {b} (N = {seq[b]}, T = {lt[b]}, gen = {gf[b]}): HIP max|err| {mx:.2e} rmse/rms {rm:.2e} PCM rmse/rms {pr:.2e} | torch-fp32 oracle "
              f"{y['max_err']:.2e} / {y['rmse_over_rms']:.2e}; PCM max diff {int(dp.max())} LSB ({int((dp > 0).sum())} of {n} differ; torch-fp32 {y['pcm_max_lsb']} LSB)")
        assert int(pre["ref_signal_len"][b]) == it["ref_signal_len"] and int(pcm_len[b]) == n == gf[b] * spec.hop_length
        if dtype == "bf16":
            checks += [(f"bf16 state rmse/rms item {b}", rm, 1.15 * yard["state"][31]), (f"bf16 pcm rmse/rms item {b}", pr, 1.15 * yard["pcm"])]
            continue
        # measured: item 1 max err 2.9e-3 (2.8 x the torch-fp32 figure, one element of a state of range 12.8) / rmse 1.25 x; item 2 1.07 x / 0.81 x
        checks += [(f"state max err item {b}", mx, 4.0 * y["max_err"] + 2e-6), (f"state rmse item {b}", rm, 2.0 * y["rmse_over_rms"] + 2e-7),
                   (f"pcm lsb item {b}", int(dp.max()), y["pcm_max_lsb"] + 1), (f"pcm share beyond 1 LSB item {b}", float((dp > 1).float().mean()), 1e-4)]
    bad = [c for c in checks if not c[1] <= c[2]]
    assert not bad, bad

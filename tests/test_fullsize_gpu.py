"""-m gpu: HIP path vs the CPU oracle at BASELINE's FULL model size (D = 1024, 22 blocks, 16 heads, N = 1600 frames,
T = 256 ids, 1037 generated frames = 265,472 samples) -- BASELINE.json configs[1] ("batch=1, 256-token utterance, fp32,
numerics vs CPU ref") and configs[2] (B = 32, bf16 acoustic + fp32 vocoder).  Inputs are exactly bench.py's
(`bench.make_inputs`, rank 0); the oracle case is item 0 of that batch.  PARITY UNPINNED against the real reference
graphs (absent offline, oracle/vv_oracle.py header): the oracle is the ground truth.

The ODE is integrated with nfe_step = 3: two Euler steps over the whole interval (sway grid t = 0, 0.293, 1; dt = 0.293,
0.707), both CFG branches each, so the state after the steps carries the two network evaluations at full weight
(with the 32-point grid the first steps have dt ~ 1e-3 and a state comparison would say nothing about the network).

Tolerances.  SURVEY.md 8(d) C2 proposed "mel rtol 1e-4/atol 1e-4, PCM +-1 LSB".  Whether fp32 arithmetic can meet that at this
size is measured, not assumed: the fixture runs the oracle TWICE on the same fp32 weights, tables and inputs -- in fp32 (plain
torch CPU) and in float64 -- and the float64 run is the ground truth.  The deviation of the fp32 oracle from it is what fp32
rounding alone costs here (22 blocks x 6 K=1024..2048 contractions, softmax over 1600 keys, CFG gain 1 + 2*2 on the branch
difference, two Euler steps with dt = 0.29 / 0.71); measured on the build machine: state max |err| 1.4e-4 / 4.7e-4 after step
0 / 1 (1.1x / 3.1x the 1e-4 + 1e-4|x| bound), rmse/rms 1.2e-5 / 2.3e-5, log-mel 1.7e-3, waveform 1.0e-5.  So the C2 figure
is not attainable by ANY fp32 implementation of this network and the assertions are stated against the measured floor:
  * fp32 state after each Euler step: max |x_hip - x_f64| <= 3 x max |x_orc32 - x_f64|, rmse <= 2 x the fp32 oracle's rmse,
    and rmse/rms <= 1e-4 outright.  The C2 ratio err / (1e-4 + 1e-4|x|) is printed for both for the record.
  * log-mel conditioning: stated on the linear mel magnitude (both sides run an fp32 1024-point DFT whose rounding noise is
    ~1e-6 of the spectral peak; bins holding only that noise sit near the 1e-5 clamp where the LOG amplifies it):
    |mel_hip - mel_f64| <= 1e-6 peak + 1e-4 mel.
  * fp32 vocoder: waveform |d| <= 3e-5 of full scale (one LSB = 3.05e-5), PCM within +-1 LSB of the float64 oracle's PCM.
  * bf16 acoustic (8-bit mantissa operands, fp32 accumulate and residual stream): every bound is 2 x the figure measured on the
    round-2 build (state rmse/rms 3.7e-3 / 6.2e-3 after the two steps, waveform 6.2e-3, batch-vs-alone 4.8e-3 / 5.2e-3).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEV = "cuda:0"
NFE = 3
N_STEPS = NFE - 1
B_HEAD = 32


@pytest.fixture(scope="module")
def full_case():
    import bench
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from oracle.vv_oracle import Oracle
    spec = ModelSpec.full()
    w = make_synthetic_weights(spec, bench.SEED)
    d, N = bench.make_inputs(spec, B_HEAD, 0, "cpu")
    assert N == 1600 and d["audio"].shape == (B_HEAD, bench.REF_SAMPLES) and d["ids"].shape == (B_HEAD, bench.TEXT_TOKENS)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(threads, 16)))
    runs = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        orc = Oracle(spec, w, nfe_step=NFE, dtype=dt)
        with torch.no_grad():
            pre = orc.preprocess(d["audio"][0], d["ids"][0], N, d["noise"][0])
            xs = [pre["noise"]]
            for st in range(N_STEPS):
                xs.append(orc.transformer_step(xs[-1], pre, st))
            wave = orc.vocoder(xs[-1][pre["ref_signal_len"]:])
        runs[name] = dict(pre=pre, xs=xs, wave=wave, pcm=orc.to_pcm(wave))
    # ground truth = float64 arithmetic on the same fp32 weights / tables / inputs; "o32" = what plain torch fp32 makes of them
    t = runs["f64"]
    return dict(spec=spec, w=w, d=d, N=N, pre=t["pre"], xs=t["xs"], wave=t["wave"], pcm=t["pcm"], o32=runs["f32"])


def _dev(d, sl):
    return {k: v[sl].contiguous().to(DEV) for k, v in d.items() if torch.is_tensor(v)}


def _run(eng, d, N, gen, n_steps=N_STEPS, want_wave=True):
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], N)
    x = d["noise"].clone()
    states = []
    for st in range(n_steps):
        eng.transformer_steps(x, pre, st, 1)
        states.append(x.clone())
    out = eng.decode(x, pre, gen, want_wave=want_wave)
    torch.cuda.synchronize()
    return pre, states, out


def test_fp32_full_size_matches_oracle(full_case):
    """configs[1]: B = 1, N = 1600, fp32 acoustic + fp32 vocoder, every stage against the oracle."""
    import bench
    from vietvoice_tts_amd.runtime import HipSynth
    c = full_case
    spec, M = c["spec"], c["spec"].n_mel
    eng = HipSynth(spec, c["w"], acoustic_dtype="fp32", nfe_step=NFE)
    pre, states, (pcm, pcm_len, wave) = _run(eng, _dev(c["d"], slice(0, 1)), c["N"], bench.GEN_FRAMES)
    # ---- preprocess graph: reference-clip log-mel, text conditioning, drop twin, frame count
    rp = c["pre"]
    assert int(pre["ref_signal_len"][0]) == rp["ref_signal_len"] == 563
    cat, catd = pre["cat_mel_text"][0].cpu(), pre["cat_mel_text_drop"][0].cpu()
    lm_h, lm_o = cat[:563, :M].double(), rp["cat_mel_text"][:563, :M].double()
    e_mel32 = float((c["o32"]["pre"]["cat_mel_text"][:563, :M].double() - lm_o).abs().max())
    e_mel = float((lm_h - lm_o).abs().max())
    mel_h, mel_o = lm_h.exp(), lm_o.exp()                       # linear mel magnitudes (the log of the clamp floor 1e-5 is -11.5)
    e_lin = float(((mel_h - mel_o).abs() / (1e-6 * float(mel_o.max()) + 1e-4 * mel_o)).max())
    e_txt = float((cat[:, M:] - rp["cat_mel_text"][:, M:]).abs().max() / rp["cat_mel_text"][:, M:].abs().max())
    e_drop = float((catd - rp["cat_mel_text_drop"]).abs().max() / rp["cat_mel_text_drop"].abs().max())
    print(f"\n[full fp32] log-mel max abs err {e_mel:.2e} (torch-fp32 oracle {e_mel32:.2e}; mel magnitudes {float(mel_o.min()):.1e}..{float(mel_o.max()):.1e}); "
          f"linear-mel worst err/(1e-6 peak + 1e-4|mel|) = {e_lin:.3f}; text cond rel {e_txt:.2e}; drop twin rel {e_drop:.2e}")
    checks = []          # collected, asserted at the end so that one run reports every stage
    # mel: both sides run an fp32 1024-point DFT whose rounding noise is ~1e-6 of the spectral peak; bins that hold only that noise
    # (the synthetic clip is 32 sinusoids) sit near the 1e-5 clamp, where the LOG amplifies it.  The bound is therefore stated on the
    # linear mel magnitude, atol = 1e-6 x peak, rtol 1e-4; the log-domain figure is printed for the record.
    checks.append(("linear mel", e_lin, 1.0))
    checks.append(("text cond", e_txt, 1e-4))
    checks.append(("drop twin", e_drop, 1e-4))
    # ---- transformer graph: state after each Euler step (both CFG branches inside)
    for st in range(N_STEPS):
        ref = c["xs"][st + 1]
        rms = float(ref.pow(2).mean().sqrt())
        stat = {}
        for who, got in (("hip", states[st][0].cpu().double()), ("orc32", c["o32"]["xs"][st + 1].double())):
            err = (got - ref).abs()
            stat[who] = (float(err.max()), float(err.pow(2).mean().sqrt()) / rms, float((err / (1e-4 + 1e-4 * ref.abs())).max()))
        print(f"[full fp32] Euler step {st} vs float64 (state range {float(ref.abs().max()):.2f}): HIP max|err| {stat['hip'][0]:.2e} rmse/rms {stat['hip'][1]:.2e} "
              f"C2 ratio {stat['hip'][2]:.2f} | torch-fp32 oracle max|err| {stat['orc32'][0]:.2e} rmse/rms {stat['orc32'][1]:.2e} C2 ratio {stat['orc32'][2]:.2f}")
        checks.append((f"state max err, Euler step {st}", stat["hip"][0], 3.0 * stat["orc32"][0]))
        checks.append((f"state rmse, Euler step {st}", stat["hip"][1], min(2.0 * stat["orc32"][1], 1e-4)))
    # ---- decode graph on the 1037 generated frames
    n = c["wave"].numel()
    assert int(pcm_len[0]) == n == bench.GEN_FRAMES * spec.hop_length
    e_w = float((wave[0, :n].cpu().double() - c["wave"]).abs().max())
    e_p = int((pcm[0, :n].cpu().int() - c["pcm"].int()).abs().max())
    e_w32 = float((c["o32"]["wave"].double() - c["wave"]).abs().max())
    print(f"[full fp32] waveform max abs err {e_w:.2e} (torch-fp32 oracle {e_w32:.2e}; peak {float(c['wave'].abs().max()):.3f}); PCM max diff {e_p} LSB")
    checks += [("waveform", e_w, 3e-5), ("pcm lsb", e_p, 1)]
    eng.close()
    bad = [c for c in checks if not c[1] <= c[2]]
    assert not bad, bad


def test_fp32_vocoder_alone_on_the_oracle_state(full_case):
    """a4 in isolation at full size: the oracle's final state through the HIP decode graph (frame slice, 4 upsample
    stages, 36 MRF resblocks, conv_post, tanh, int16) vs the oracle's vocoder on the same frames."""
    import bench
    from vietvoice_tts_amd.runtime import HipSynth
    c = full_case
    eng = HipSynth(c["spec"], c["w"], acoustic_dtype="bf16", nfe_step=NFE)
    x = c["xs"][-1].float().unsqueeze(0).to(DEV)          # the float64 oracle's final state, rounded once to fp32
    pre = {"ref_signal_len": torch.tensor([563], dtype=torch.int32, device=DEV), "seq_len": torch.tensor([c["N"]], dtype=torch.int32, device=DEV)}
    eng.set_option("voc_x3", 0)                             # products on v_mfma_f32_32x32x2_f32 (the fp32 context's default)
    pcm, pcm_len, wave = eng.decode(x, pre, bench.GEN_FRAMES, want_wave=True)
    torch.cuda.synchronize()
    n = c["wave"].numel()
    e_w = float((wave[0, :n].cpu().double() - c["wave"]).abs().max())
    r_w = float((wave[0, :n].cpu().double() - c["wave"]).pow(2).mean().sqrt())
    diff = (pcm[0, :n].cpu().int() - c["pcm"].int()).abs()
    print(f"\n[full vocoder] waveform max abs err {e_w:.2e} (rms {r_w:.2e}); PCM max diff {int(diff.max())} LSB on {int((diff > 0).sum())} of {n} samples")
    assert int(pcm_len[0]) == n and e_w < 1e-5 and int(diff.max()) <= 1
    # x3: the same graph with every conv product taken as an exact 3-way bf16 split on the bf16 matrix pipe (the bf16 context's
    # default).  Same tolerance against the float64 oracle as the f32 instruction; the two fp32-class results differ from each
    # other by accumulation rounding only.
    eng.set_option("voc_x3", 1)
    pcm3, len3, wave3 = eng.decode(x, pre, bench.GEN_FRAMES, want_wave=True)
    torch.cuda.synchronize()
    e3 = float((wave3[0, :n].cpu().double() - c["wave"]).abs().max())
    r3 = float((wave3[0, :n].cpu().double() - c["wave"]).pow(2).mean().sqrt())
    diff3 = (pcm3[0, :n].cpu().int() - c["pcm"].int()).abs()
    print(f"[full vocoder x3] waveform max abs err {e3:.2e} (rms {r3:.2e}); PCM max diff {int(diff3.max())} LSB on {int((diff3 > 0).sum())} of {n} samples; "
          f"x3 vs f32-MFMA waveform max diff {float((wave3 - wave).abs().max()):.2e}")
    assert int(len3[0]) == n and e3 < 1e-5 and int(diff3.max()) <= 1 and r3 <= 2.0 * r_w + 1e-7
    eng.prof_enable(True)
    eng.decode(x, pre, bench.GEN_FRAMES)
    pr = eng.prof_collect()["voc_conv"]
    print(f"[full vocoder x3] {pr['launches']} conv launches, {pr['ms']:.3f} ms (B = 1)")
    eng.prof_enable(False)
    eng.set_option("voc_x3", 0)
    # K12: the fused MRF pairs (stages with C = 64 and 32, 18 of the 36 pairs) against two launches per pair: bit-identical
    eng.set_option("fuse_mrf", 0)
    pcm_u, len_u, wave_u = eng.decode(x, pre, bench.GEN_FRAMES, want_wave=True)
    torch.cuda.synchronize()
    assert torch.equal(pcm_u, pcm) and torch.equal(wave_u, wave) and torch.equal(len_u, pcm_len)
    eng.prof_enable(True)
    for fuse in (1, 0):
        eng.set_option("fuse_mrf", fuse)
        eng.decode(x, pre, bench.GEN_FRAMES)
        pr = eng.prof_collect()["voc_conv"]
        print(f"[full vocoder] fuse_mrf={fuse}: {pr['launches']} conv launches, {pr['ms']:.3f} ms (B = 1)")
    eng.prof_enable(False)
    eng.close()


def test_bf16_full_size_close_to_oracle_and_b32_properties(full_case):
    """configs[2] shapes.  (a) B = 1 bf16 acoustic vs the fp32 oracle: state RMSE per Euler step; (b) the headline batch
    B = 32 (bench inputs): item 0 inside the batch equals item 0 alone (rows are packed and every kernel is row- or
    sequence-local; the two batch sizes differ only in which GEMM tiling runs), the whole batch is finite and full length;
    (c) item 0 of the B = 32 batch against the oracle (same tolerance as (a))."""
    import bench
    from vietvoice_tts_amd.runtime import HipSynth
    c = full_case
    spec = c["spec"]
    eng = HipSynth(spec, c["w"], acoustic_dtype="bf16", nfe_step=NFE)
    _, st1, (pcm1, len1, wave1) = _run(eng, _dev(c["d"], slice(0, 1)), c["N"], bench.GEN_FRAMES)
    for st in range(N_STEPS):
        got, ref = st1[st][0].cpu().double(), c["xs"][st + 1]
        rmse = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"\n[full bf16 B=1] Euler step {st}: state rmse/rms {rmse:.3e}, max abs err {float((got - ref).abs().max()):.3e}")
        # measured 3.7e-3 / 6.2e-3 after step 0 / 1 (round 2, same inputs): bound = 2 x measured, so a regression that doubles the
        # bf16 error fails
        assert rmse < (7.5e-3, 1.25e-2)[st], (st, rmse)
    n = c["wave"].numel()
    wr = float((wave1[0, :n].cpu().double() - c["wave"]).pow(2).mean().sqrt() / c["wave"].pow(2).mean().sqrt())
    print(f"[full bf16 B=1] waveform rmse/rms vs the fp32 oracle {wr:.3e}")
    assert wr < 1.25e-2                      # measured 6.2e-3; 2 x
    # ---- the headline batch
    _, st32, (pcm32, len32, wave32) = _run(eng, _dev(c["d"], slice(0, B_HEAD)), c["N"], bench.GEN_FRAMES)
    assert int(len32.sum()) == B_HEAD * n and bool((len32 == n).all())             # what bench.py asserts, per item
    x32 = st32[-1]
    assert bool(torch.isfinite(x32).all()) and bool(torch.isfinite(wave32).all())
    assert float(pcm32.float().abs().amax(dim=1).min()) > 0                        # no silent item
    ref0 = c["xs"][-1]
    rms0 = float(ref0.pow(2).mean().sqrt())
    d0 = float((x32[0] - st1[-1][0]).double().pow(2).mean().sqrt()) / rms0
    rmse32 = float((x32[0].cpu().double() - ref0).pow(2).mean().sqrt()) / rms0
    wd = float((wave32[0, :n] - wave1[0, :n]).double().pow(2).mean().sqrt() / wave1[0, :n].double().pow(2).mean().sqrt())
    print(f"[full bf16 B=32] item 0 in the batch vs alone: state rmse/rms {d0:.3e}, waveform rmse/rms {wd:.3e}; vs float64 oracle rmse/rms {rmse32:.3e}")
    # M = 102,400 rows take the persistent 256x256 GEMM, M = 3,200 the 128x128 kernel: different bf16 rounding points, same tolerance class
    assert d0 < 1e-2 and wd < 1.05e-2                                              # measured 4.8e-3 / 5.2e-3; 2 x
    assert rmse32 < 1.25e-2                                                        # measured 6.2e-3; 2 x
    # items differ (different clips / ids / noise): the batch is not one utterance repeated
    assert float((x32[1] - x32[0]).abs().max()) > 1e-2
    # the split-K tail option of the gate-store GEMMs (off by default since round 4; 2 = FF2, 1 adds the out-projection; this batch has
    # 1,600 tiles = 6.25 rounds, so a requested tail is taken): rows of the last panels sum their K parts in fp32 in another order than
    # the MFMA accumulator does -- same tolerance class -- and the rows that are not in the tail of any GEMM differ only through
    # attention over (identical) own-sequence rows: item 0 is rows [0, N) of the conditional branch, far below row0 = 98,304, and
    # must be bit-identical.
    eng.set_option("split_k_tail", 0)
    _, st32n, (_, _, wave32n) = _run(eng, _dev(c["d"], slice(0, B_HEAD)), c["N"], bench.GEN_FRAMES)
    assert torch.equal(wave32n[0], wave32[0])
    assert torch.equal(st32n[-1], x32)                         # the default is mode 0, and runs are deterministic
    for mode in (2, 1):
        eng.set_option("split_k_tail", mode)
        _, st32m, (_, _, wave32m) = _run(eng, _dev(c["d"], slice(0, B_HEAD)), c["N"], bench.GEN_FRAMES)
        dt = float((st32m[-1] - st32n[-1]).double().pow(2).mean().sqrt()) / rms0
        dl = float((st32m[-1][-1] - st32n[-1][-1]).double().pow(2).mean().sqrt()) / rms0
        print(f"[full bf16 B=32] split-K tail mode {mode} vs off: state rmse/rms whole batch {dt:.3e}, last item (in the tail) {dl:.3e}")
        assert 0 < dt < 5e-3 and 0 < dl < 1e-2, "the tail is taken at this shape (> 0) and stays in the bf16 tolerance class"
        assert torch.equal(wave32m[0], wave32[0])
    eng.set_option("split_k_tail", 0)
    eng.close()


def test_fp32_full_size_ragged_batch_matches_oracle(full_case):
    """configs[3] semantics at FULL model size: a ragged batch (three utterances of different reference-clip, text and frame
    lengths, packed rows, per-item masks in attention / pos-conv / text conv, bucketed vocoder) through the fp32 path, one Euler
    step over the whole interval, every item against the fp32 oracle run on that item alone.  Tolerance as derived in the module
    docstring: state rmse/rms <= 1e-4 (fp32 floor measured 1.3e-5 for one step), max error <= 1e-3 of the state range, PCM +-1 LSB
    on >= 99.9 % of the samples (two fp32 implementations are compared here, each ~1e-5 from float64)."""
    import bench
    from vietvoice_tts_amd.runtime import HipSynth
    from oracle.vv_oracle import Oracle
    c = full_case
    spec = c["spec"]
    g = torch.Generator().manual_seed(77)
    la, lt, gf = [144000, 256 * 330 + 100, 256 * 150], [256, 140, 64], [1037, 520, 260]
    B = len(la)
    audio = torch.zeros(B, max(la), dtype=torch.int16)
    ids = torch.zeros(B, max(lt), dtype=torch.int32)
    for b in range(B):
        audio[b, : la[b]] = bench.synth_reference_clip(100 + b, la[b])
        ids[b, : lt[b]] = torch.randint(1, spec.vocab_size, (lt[b],), generator=g, dtype=torch.int32)
    seq = [la[b] // spec.hop_length + 1 + gf[b] for b in range(B)]
    N = max(seq)
    noise = torch.randn(B, N, spec.n_mel, generator=g)
    orc = Oracle(spec, c["w"], nfe_step=2)               # one Euler step, dt = 1
    eng = HipSynth(spec, c["w"], acoustic_dtype="fp32", nfe_step=2)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    pre = eng.preprocess(audio.to(DEV), i32(la), ids.to(DEV), i32(lt), i32(seq), N)
    x = noise.to(DEV).clone()
    eng.transformer_steps(x, pre, 0, 1)
    pcm, pcm_len = eng.decode_bucketed(x, pre, gf, pad_frac=0.10, min_units=1)
    torch.cuda.synchronize()
    for b in range(B):
        with torch.no_grad():
            p = orc.preprocess(audio[b, : la[b]], ids[b, : lt[b]], seq[b], noise[b, : seq[b]])
            xr = orc.transformer_step(p["noise"], p, 0)
            pr = orc.to_pcm(orc.vocoder(xr[p["ref_signal_len"]:]))
        got = x[b, : seq[b]].cpu()
        err = (got - xr).abs()
        rm = float(err.pow(2).mean().sqrt() / xr.pow(2).mean().sqrt())
        n = pr.numel()
        d = (pcm[b, :n].cpu().int() - pr.int()).abs()
        fr, mb = divmod(int(err.argmax()), spec.n_mel)
        print(f"\n[full fp32 ragged] item {b} (N={seq[b]}, T={lt[b]}, gen={gf[b]}): state rmse/rms {rm:.2e}, max {float(err.max()):.2e} (frame {fr} of "
              f"{seq[b]}, ref frames {p['ref_signal_len']}, bin {mb}; p99.9 {float(err.flatten().kthvalue(int(err.numel() * 0.999)).values):.2e}) of range "
              f"{float(xr.abs().max()):.1f}; PCM max diff {int(d.max())} LSB on {int((d > 0).sum())} of {n}")
        assert int(pre["ref_signal_len"][b]) == p["ref_signal_len"] and int(pcm_len[b]) == n == gf[b] * spec.hop_length
        assert rm <= 1e-4 and float(err.max()) <= 1e-3 * float(xr.abs().max())
        assert int(d.max()) <= 2 and float((d > 1).float().mean()) < 1e-3
    eng.close()

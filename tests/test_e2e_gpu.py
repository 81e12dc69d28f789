"""-m gpu: the three stages end to end through the C ABI vs the CPU oracle (tiny dims, same seeded
weights / inputs / explicit noise).  PARITY UNPINNED against the real reference graphs (absent,
see oracle/vv_oracle.py); the oracle is the ground truth here.

Stated tolerances (north_star: "within a stated fp32 mel/waveform tolerance"):
  fp32 path : log-mel conditioning 2e-3 abs; state after all Euler steps 1e-3 of its range;
              waveform 2e-4 abs (full scale 1.0); PCM within +-2 LSB of the oracle's int16.
  bf16 path : state RMSE < 2% of the state RMS (bf16 operands, fp32 accumulate / residual stream).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def make_batch(spec, lens_audio, lens_text, gen_frames, seed=0):
    g = torch.Generator().manual_seed(seed)
    B = len(lens_audio)
    S = max(lens_audio)
    T = max(lens_text)
    audio = torch.zeros(B, S, dtype=torch.int16)
    ids = torch.zeros(B, T, dtype=torch.int32)
    for b in range(B):
        audio[b, : lens_audio[b]] = (torch.randn(lens_audio[b], generator=g) * 3000).to(torch.int16)
        ids[b, : lens_text[b]] = torch.randint(0, spec.vocab_size, (lens_text[b],), generator=g, dtype=torch.int32)
    seq = [la // spec.hop_length + 1 + gf for la, gf in zip(lens_audio, gen_frames)]
    N = max(seq)
    noise = torch.randn(B, N, spec.n_mel, generator=g)
    return dict(audio=audio, audio_len=torch.tensor(lens_audio, dtype=torch.int32), ids=ids,
                text_len=torch.tensor(lens_text, dtype=torch.int32), seq_len=torch.tensor(seq, dtype=torch.int32), N=N,
                noise=noise, t_gen_max=max(gen_frames))


def run_oracle(orc, batch, n_steps):
    outs = []
    for b in range(batch["audio"].shape[0]):
        la, lt, sl = int(batch["audio_len"][b]), int(batch["text_len"][b]), int(batch["seq_len"][b])
        pre = orc.preprocess(batch["audio"][b, :la], batch["ids"][b, :lt], sl, batch["noise"][b, :sl])
        x = pre["noise"]
        for st in range(n_steps):
            x = orc.transformer_step(x, pre, st)
        wave = orc.vocoder(x[pre["ref_signal_len"]:])
        outs.append(dict(pre=pre, x=x, wave=wave, pcm=orc.to_pcm(wave)))
    return outs


def run_hip(eng, batch, n_steps, want_wave=True):
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x = d["noise"].clone()
    eng.transformer_steps(x, pre, 0, n_steps)
    pcm, pcm_len, wave = eng.decode(x, pre, d["t_gen_max"], want_wave=True)
    torch.cuda.synchronize()
    return pre, x.cpu(), pcm.cpu(), pcm_len.cpu(), wave.cpu()


@pytest.mark.parametrize("lens", [
    dict(a=[256 * 20, 256 * 20], t=[30, 30], g=[24, 24]),                 # uniform batch
    dict(a=[256 * 20, 256 * 12 + 100, 256 * 30], t=[30, 11, 47], g=[24, 9, 40]),   # ragged batch: masks everywhere
])
def test_fp32_pipeline_matches_oracle(hip_tiny, tiny_setup, lens):
    spec, _, orc = tiny_setup
    eng = hip_tiny["f32"]
    batch = make_batch(spec, lens["a"], lens["t"], lens["g"], seed=len(lens["a"]))
    n_steps = 7
    ref = run_oracle(orc, batch, n_steps)
    pre, x, pcm, pcm_len, wave = run_hip(eng, batch, n_steps)
    M = spec.n_mel
    for b, r in enumerate(ref):
        sl = int(batch["seq_len"][b])
        rl = r["pre"]["ref_signal_len"]
        assert int(pre["ref_signal_len"][b]) == rl
        cat = pre["cat_mel_text"][b, :sl].cpu()
        catd = pre["cat_mel_text_drop"][b, :sl].cpu()
        assert float((cat[:, :M] - r["pre"]["cat_mel_text"][:, :M]).abs().max()) < 2e-3
        assert float((cat[:, M:] - r["pre"]["cat_mel_text"][:, M:]).abs().max()) < 2e-3 * float(r["pre"]["cat_mel_text"][:, M:].abs().max())
        assert float((catd - r["pre"]["cat_mel_text_drop"]).abs().max()) < 2e-3 * float(r["pre"]["cat_mel_text_drop"].abs().max())
        err_x = float((x[b, :sl] - r["x"]).abs().max()) / float(r["x"].abs().max())
        assert err_x < 1e-3, err_x
        n = r["wave"].numel()
        assert int(pcm_len[b]) == n
        assert float((wave[b, :n] - r["wave"]).abs().max()) < 2e-4
        assert int((pcm[b, :n].int() - r["pcm"].int()).abs().max()) <= 2


def test_bf16_pipeline_close_to_oracle(hip_tiny, tiny_setup):
    spec, _, orc = tiny_setup
    eng = hip_tiny["bf16"]
    batch = make_batch(spec, [256 * 20, 256 * 14], [30, 21], [24, 17], seed=5)
    n_steps = 7
    ref = run_oracle(orc, batch, n_steps)
    pre, x, pcm, pcm_len, wave = run_hip(eng, batch, n_steps)
    for b, r in enumerate(ref):
        sl = int(batch["seq_len"][b])
        d = x[b, :sl] - r["x"]
        rmse = float(d.pow(2).mean().sqrt() / r["x"].pow(2).mean().sqrt())
        assert rmse < 2e-2, rmse
        n = r["wave"].numel()
        assert int(pcm_len[b]) == n
        wr = float((wave[b, :n] - r["wave"]).pow(2).mean().sqrt() / r["wave"].pow(2).mean().sqrt())
        assert wr < 0.1, wr


def test_step_splitting_is_exact(hip_tiny, tiny_setup):
    """Size-independent property: 7 steps in one call == 3 + 4 steps in two calls (state lives in HBM)."""
    spec, _, _ = tiny_setup
    eng = hip_tiny["f32"]
    batch = make_batch(spec, [256 * 16], [20], [12], seed=8)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x1 = d["noise"].clone()
    eng.transformer_steps(x1, pre, 0, 7)
    x2 = d["noise"].clone()
    eng.transformer_steps(x2, pre, 0, 3)
    eng.transformer_steps(x2, pre, 3, 4)
    torch.cuda.synchronize()
    assert torch.equal(x1, x2)


def test_host_lengths_form_is_identical_and_graph_capturable(hip_tiny, tiny_setup):
    """vv_transformer_steps_h (lengths also given on the host): no read-back and no stream synchronisation inside the call, the
    packed-row tables are built on the device.  Same state bit for bit as the read-back form, on a ragged batch; and with the
    context arena already sized the whole Euler-step loop can be captured into a hipGraph and replayed."""
    spec, _, _ = tiny_setup
    eng = hip_tiny["f32"]
    batch = make_batch(spec, [256 * 20, 256 * 12 + 100, 256 * 30], [30, 11, 47], [24, 9, 40], seed=19)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x_dev = d["noise"].clone()
    eng.transformer_steps(x_dev, pre, 0, 5)                                   # lengths read back from the device
    lens = [int(v) for v in batch["seq_len"]]
    x_host = d["noise"].clone()
    eng.transformer_steps(x_host, pre, 0, 5, seq_len_host=lens)               # lengths from the host: sync-free
    torch.cuda.synchronize()
    assert torch.equal(x_dev, x_host)
    x_g = d["noise"].clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.transformer_steps(x_g, pre, 0, 5, seq_len_host=lens)
    x_g.copy_(d["noise"])
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x_g, x_dev)
    with pytest.raises(RuntimeError, match="seq_len"):                        # host lengths are validated like device lengths
        eng.transformer_steps(d["noise"].clone(), pre, 0, 1, seq_len_host=[lens[0], 0, lens[2]])


def test_batch_equals_singles(hip_tiny, tiny_setup):
    """Batched ragged synthesis must equal each utterance synthesised alone (key-padding / edge masks)."""
    spec, _, _ = tiny_setup
    eng = hip_tiny["f32"]
    la, lt, gf = [256 * 20, 256 * 11 + 50], [25, 13], [20, 8]
    batch = make_batch(spec, la, lt, gf, seed=12)
    _, xb, pcmb, lenb, waveb = run_hip(eng, batch, 4)
    for b in range(2):
        sl = int(batch["seq_len"][b])
        single = dict(audio=batch["audio"][b:b + 1, : la[b]].contiguous(), audio_len=batch["audio_len"][b:b + 1],
                      ids=batch["ids"][b:b + 1, : lt[b]].contiguous(), text_len=batch["text_len"][b:b + 1],
                      seq_len=batch["seq_len"][b:b + 1], N=sl, noise=batch["noise"][b:b + 1, :sl].contiguous(), t_gen_max=gf[b])
        _, xs, pcms, lens_, waves = run_hip(eng, single, 4)
        n = int(lens_[0])
        assert int(lenb[b]) == n
        assert float((xb[b, :sl] - xs[0]).abs().max()) < 5e-5 * float(xs[0].abs().max()) + 1e-6
        assert int((pcmb[b, :n].int() - pcms[0, :n].int()).abs().max()) <= 1


def test_graph_captured_decode_matches_eager(hip_tiny, tiny_setup):
    """Config 5: the vocoder step captured into a hipGraph replays bit-exactly for new inputs."""
    spec, _, _ = tiny_setup
    eng = hip_tiny["f32"]
    batch = make_batch(spec, [256 * 16, 256 * 16], [20, 14], [12, 9], seed=21)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x = d["noise"].clone()
    eng.transformer_steps(x, pre, 0, 3)
    graphed = eng.capture_decode(2, d["N"], d["t_gen_max"])
    for scale in (1.0, 0.5):
        xin = x * scale
        pcm_e, len_e = eng.decode(xin, pre, d["t_gen_max"])
        pcm_g, len_g = graphed(xin, pre["ref_signal_len"], pre["seq_len"])
        torch.cuda.synchronize()
        assert torch.equal(len_e, len_g) and torch.equal(pcm_e, pcm_g)


def test_captured_decode_survives_workspace_growth(tiny_setup):
    """a captured vocoder graph must stay valid when a LATER, larger call reallocates the context arena.
    Capture at a small shape, force the arena to move (vv_ws_generation changes) with a much larger batch, then replay the
    old graph with fresh inputs and compare with the eager decode, bit for bit."""
    from vietvoice_tts_amd.runtime import HipSynth
    spec, w, _ = tiny_setup
    eng = HipSynth(spec, w, acoustic_dtype="fp32", nfe_step=4)          # its own context: the arena starts small
    small = make_batch(spec, [256 * 16, 256 * 16], [20, 14], [12, 9], seed=41)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in small.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x = d["noise"].clone()
    eng.transformer_steps(x, pre, 0, 2)
    graphed = eng.capture_decode(2, d["N"], d["t_gen_max"])
    pcm_g0 = graphed(x, pre["ref_signal_len"], pre["seq_len"])[0].clone()
    gen0 = int(eng.lib.vv_ws_generation(eng.ctx))
    big = make_batch(spec, [256 * 40] * 6, [40] * 6, [300] * 6, seed=42)                 # ~25x the decode workspace
    db = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in big.items()}
    eng.synthesize_batch(db["audio"], db["audio_len"], db["ids"], db["text_len"], db["seq_len"], db["N"], db["noise"], db["t_gen_max"], n_steps=1)
    torch.cuda.synchronize()
    assert int(eng.lib.vv_ws_generation(eng.ctx)) > gen0, "the larger call was meant to move the arena"
    for scale in (1.0, -0.7):
        xin = x * scale
        pcm_g, len_g = graphed(xin, pre["ref_signal_len"], pre["seq_len"])
        pcm_g = pcm_g.clone()
        pcm_e, len_e = eng.decode(xin, pre, d["t_gen_max"])
        torch.cuda.synchronize()
        assert torch.equal(len_e, len_g) and torch.equal(pcm_e, pcm_g)
    assert torch.equal(pcm_g0, graphed(x, pre["ref_signal_len"], pre["seq_len"])[0])
    eng.close()


def test_bucketed_decode_equals_one_batch(hip_tiny, tiny_setup):
    """Ragged batches decode in length buckets (padded vocoder planes): same PCM as the single padded decode, bit for bit."""
    spec, _, _ = tiny_setup
    eng = hip_tiny["f32"]
    gf = [40, 9, 38, 8, 10, 41, 39, 11, 37]
    batch = make_batch(spec, [256 * 12] * len(gf), [20] * len(gf), gf, seed=31)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x = d["noise"].clone()
    eng.transformer_steps(x, pre, 0, 2)
    pcm_1, len_1 = eng.decode(x, pre, max(gf))
    pcm_b, len_b = eng.decode_bucketed(x, pre, gf, pad_frac=0.10, min_units=4)
    torch.cuda.synchronize()
    assert torch.equal(len_1, len_b)
    for b in range(len(gf)):
        n = int(len_1[b])
        assert n == gf[b] * spec.hop_length and torch.equal(pcm_1[b, :n], pcm_b[b, :n])


def test_bf16_pipeline_through_the_persistent_gemm(hip_tiny):
    """'small' preset (D = 256, 3 blocks) on a ragged batch whose packed row count passes 4096, so every block GEMM takes
    the persistent 256x256 ping-pong kernel (QKV rope with the per-row position table, gate-store, GELU store with the
    relaxed first-K-tile wait), attention and pos-conv run on packed rows of four different lengths, and the two-delta /
    keep_x LayerNorm protocol carries the residual stream -- against the fp32 CPU oracle, bf16 tolerance."""
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    from oracle.vv_oracle import Oracle
    spec = ModelSpec.small()
    w = make_synthetic_weights(spec, seed=4242)
    orc = Oracle(spec, w, nfe_step=4)
    eng = HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=4)
    eng.set_option("pp_min_tiles", 0)      # a small model's GEMMs have far fewer 256-tiles than the chip has CUs: take the persistent kernel anyway
    eng.set_option("lanes", 1)             # ... with the whole batch in one launch (two lanes of ~2,400 rows would fall under 4096)
    la = [256 * 60, 256 * 45 + 17, 256 * 70, 256 * 52]
    gf = [560, 520, 470, 540]
    batch = make_batch(spec, la, [40, 33, 47, 38], gf, seed=77)
    assert 2 * int(batch["seq_len"].sum()) >= 4096
    n_steps = 3
    ref = run_oracle(orc, batch, n_steps)
    pre, x, pcm, pcm_len, wave = run_hip(eng, batch, n_steps)
    for b, r in enumerate(ref):
        sl = int(batch["seq_len"][b])
        d = x[b, :sl] - r["x"]
        rmse = float(d.pow(2).mean().sqrt() / r["x"].pow(2).mean().sqrt())
        assert rmse < 2e-2, (b, rmse)
        n = r["wave"].numel()
        assert int(pcm_len[b]) == n
    eng.close()


def test_bf16_item_in_a_batch_equals_the_item_alone_across_gemm_kernels():
    """Round 4: the reference synthesises every unit as an independent B = 1 call (/root/reference/vietvoicetts/core/tts_engine.py:47,121),
    so an item inside a batch must equal the same item alone -- in bf16 too, although the packed batch (>= 4096 rows) takes the
    persistent 256 x 256 GEMM and the item alone the 128 x 128 kernel.  Both kernels contract K in the same order with the same MFMA,
    start their accumulators at the bias and share one epilogue arithmetic (vv_gemm.hip: act_pair / rope_pair), every other kernel is
    row- or sequence-local: state and PCM are BIT-IDENTICAL ('small' preset: D = 256, 3 blocks; ragged batch of four)."""
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    spec = ModelSpec.small()
    w = make_synthetic_weights(spec, seed=4242)
    eng = HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=6)
    eng.set_option("pp_min_tiles", 0)      # the packed batch on the persistent kernel although a small model has few 256-tiles
    eng.set_option("lanes", 1)
    la = [256 * 60, 256 * 45 + 17, 256 * 70, 256 * 52]
    lt = [40, 33, 47, 38]
    gf = [560, 520, 470, 540]
    batch = make_batch(spec, la, lt, gf, seed=78)
    assert 2 * int(batch["seq_len"].sum()) >= 4096 and 2 * int(batch["seq_len"].max()) < 4096
    pre, x, pcm, pcm_len, _w = run_hip(eng, batch, 5, want_wave=False)
    for b in range(4):
        sl = int(batch["seq_len"][b])
        one = dict(audio=batch["audio"][b:b + 1, : la[b]].contiguous(), audio_len=batch["audio_len"][b:b + 1], ids=batch["ids"][b:b + 1, : lt[b]].contiguous(),
                   text_len=batch["text_len"][b:b + 1], seq_len=batch["seq_len"][b:b + 1], N=sl, noise=batch["noise"][b:b + 1, :sl].contiguous(), t_gen_max=gf[b])
        _p, x1, pcm1, pcm_len1, _w1 = run_hip(eng, one, 5, want_wave=False)
        n = int(pcm_len1[0])
        assert n == int(pcm_len[b]) == gf[b] * spec.hop_length
        assert torch.equal(x1[0], x[b, :sl]), (b, float((x1[0] - x[b, :sl]).abs().max()))
        assert torch.equal(pcm1[0, :n], pcm[b, :n]), b
    eng.close()


def test_captured_step_loop_and_decode_equal_eager():
    """all Euler steps (vv_transformer_steps_into) + the decode (vv_decode_into) captured into ONE hipGraph
    (runtime.GraphedSteps: static buffers and one workspace block owned by the graph object) and replayed: state and PCM bit-identical
    to the eager calls, at 'small' in bf16 and fp32, on a ragged batch; a second replay with other inputs of the same lengths is
    right too (nothing of the first call is baked in), and the context arena may be reallocated in between."""
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    spec = ModelSpec.small()
    w = make_synthetic_weights(spec, seed=4242)
    for dt in ("bf16", "fp32"):
        eng = HipSynth(spec, w, acoustic_dtype=dt, nfe_step=6)
        la, lt, gf = [256 * 20, 256 * 12 + 100, 256 * 30], [30, 11, 47], [24, 9, 40]
        graphs = None
        for seed in (5, 6):
            batch = make_batch(spec, la, lt, gf, seed=seed)
            d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
            lens = [int(v) for v in batch["seq_len"]]
            pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"], seq_len_host=lens)
            x_e = d["noise"].clone()
            eng.transformer_steps(x_e, pre, 0, eng.n_steps)
            pcm_e, len_e = eng.decode(x_e, pre, batch["t_gen_max"])
            if graphs is None:
                graphs = eng.capture_steps(len(la), d["N"], lens, batch["t_gen_max"])
                # grow the context arena after the capture: the graph must not point into it
                big = make_batch(spec, [256 * 90] * 4, [60] * 4, [300] * 4, seed=1)
                run_hip(eng, big, 1, want_wave=False)
            x_g, pcm_g, len_g = graphs(d["noise"], pre)
            torch.cuda.synchronize()
            assert torch.equal(x_g, x_e), (dt, seed)
            assert torch.equal(len_g, len_e) and torch.equal(pcm_g, pcm_e), (dt, seed)
        with pytest.raises(RuntimeError, match="workspace block too small"):
            eng._check(eng.lib.vv_transformer_steps_into(eng.ctx, 3, d["N"], d["seq_len"].data_ptr(), graphs._host, x_e.data_ptr(), pre["cat_mel_text"].data_ptr(),
                                                         pre["cat_mel_text_drop"].data_ptr(), eng.rope[0].data_ptr(), eng.rope[1].data_ptr(), eng.rope[2].data_ptr(),
                                                         eng.rope[3].data_ptr(), 0, 1, graphs.ws.data_ptr(), 4096, torch.cuda.current_stream().cuda_stream))
        # the capture bakes in pointers to the context's time-grid tables; set_nfe frees them -> a replay must refuse
        assert not graphs.stale()
        eng.set_nfe(8)
        assert graphs.stale()
        with pytest.raises(RuntimeError, match="stale"):
            graphs(d["noise"], pre)
        fresh = eng.capture_steps(len(la), d["N"], lens, batch["t_gen_max"])          # capturing again on the new grid works
        x_e = d["noise"].clone()
        eng.transformer_steps(x_e, pre, 0, eng.n_steps)
        x_g, _pcm, _len = fresh(d["noise"], pre)
        torch.cuda.synchronize()
        assert torch.equal(x_g, x_e)
        eng.close()


def test_two_lanes_equal_one_lane_bit_for_bit():
    """The Euler steps of a batch as two half batches on two HIP streams (option "lanes" 2: lane 1 on a context-owned side stream
    forked from and joined to the caller's stream inside the call) against the one-lane call: every item's state and PCM bit-identical,
    in bf16 and fp32, on a single item (then the two CFG branches are the lanes: conditional and unconditional rows of the same packed
    buffers, fork / join once per step) and on ragged batches of 2 / 3 / 5 items (cut at the item boundary closest to half of the rows), with the lengths
    read back from the device and handed over on the host, split across two calls, and captured into ONE hipGraph (the side stream
    joins the capture through the fork event).  Each lane is a complete sub-problem, so this is the batch-invariance property again."""
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    spec = ModelSpec.small()
    w = make_synthetic_weights(spec, seed=77)
    cases = [([256 * 26], [41], [33]),                 # ONE item: the lanes are its two CFG branches, forked and joined once per step
             ([256 * 20, 256 * 31], [30, 47], [24, 40]), ([256 * 20, 256 * 12 + 100, 256 * 30], [30, 11, 47], [24, 9, 40]),
             ([256 * 9, 256 * 40, 256 * 12, 256 * 25, 256 * 6], [8, 50, 20, 33, 5], [10, 44, 17, 30, 7])]
    for dt in ("bf16", "fp32"):
        eng = HipSynth(spec, w, acoustic_dtype=dt, nfe_step=6)
        for la, lt, gf in cases:
            batch = make_batch(spec, la, lt, gf, seed=len(la))
            d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
            lens = [int(v) for v in batch["seq_len"]]
            pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"], seq_len_host=lens)
            eng.set_option("lanes", 1)
            x1 = d["noise"].clone()
            eng.transformer_steps(x1, pre, 0, eng.n_steps)
            pcm1, len1 = eng.decode(x1, pre, batch["t_gen_max"])
            eng.set_option("lanes", 2)
            x2 = d["noise"].clone()
            eng.transformer_steps(x2, pre, 0, 2)                    # split across calls, host lengths
            eng.transformer_steps(x2, pre, 2, eng.n_steps - 2)
            pre_dev = dict(pre)
            pre_dev.pop("seq_len_host", None)                       # lengths read back from the device
            x3 = d["noise"].clone()
            eng.transformer_steps(x3, pre_dev, 0, eng.n_steps)
            torch.cuda.synchronize()
            assert torch.equal(x2, x1) and torch.equal(x3, x1), (dt, len(la))
            pcm2, len2 = eng.decode(x2, pre, batch["t_gen_max"])
            assert torch.equal(pcm2, pcm1) and torch.equal(len2, len1)
            graph = eng.capture_steps(len(la), d["N"], lens, batch["t_gen_max"])      # captured with two lanes
            xg, pcmg, leng = graph(d["noise"], pre)
            torch.cuda.synchronize()
            assert torch.equal(xg, x1) and torch.equal(pcmg, pcm1) and torch.equal(leng, len1), (dt, len(la), "graph")
        # the table forms of the rope (standard tables read instead of computed; then gathered per packed row): the branch views offset
        # the per-row tables by their first row, the item lanes build their own
        for rope_rows in (0, 1):
            eng.set_rope_theta(0.0)
            eng.set_option("rope_rows", rope_rows)
            for la, lt, gf in cases[:3]:
                batch = make_batch(spec, la, lt, gf, seed=10 + len(la))
                d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
                lens = [int(v) for v in batch["seq_len"]]
                pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"], seq_len_host=lens)
                xs = []
                for lanes in (1, 2):
                    eng.set_option("lanes", lanes)
                    x = d["noise"].clone()
                    eng.transformer_steps(x, pre, 0, 3)
                    xs.append(x)
                torch.cuda.synchronize()
                assert torch.equal(xs[0], xs[1]), (dt, rope_rows, len(la))
        eng.set_option("lanes", 0)
        eng.close()


def test_full_size_model_properties():
    """BASELINE's full architecture (22 x 1024, 336 M parameters, bf16 acoustic), N = 1600-frame utterances: size-independent
    properties, no oracle run needed.  (a) splitting the Euler steps across calls is bit-exact; (b) an utterance synthesised
    inside a ragged batch equals the same utterance alone (different GEMM tilings / packed-row offsets: bf16 tolerance);
    (c) the PCM is finite, full length, and not silent."""
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    spec = ModelSpec.full()
    eng = HipSynth(spec, make_synthetic_weights(spec, seed=9527), acoustic_dtype="bf16", nfe_step=4)
    la, lt, gf = [144000, 256 * 300 + 40], [256, 120], [1037, 700]
    batch = make_batch(spec, la, lt, gf, seed=3)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    pre = eng.preprocess(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], d["N"])
    x_once = d["noise"].clone()
    eng.transformer_steps(x_once, pre, 0, 3)
    x_split = d["noise"].clone()
    eng.transformer_steps(x_split, pre, 0, 1)
    eng.transformer_steps(x_split, pre, 1, 2)
    torch.cuda.synchronize()
    assert torch.equal(x_once, x_split)
    pcm, pcm_len = eng.decode(x_once, pre, max(gf))
    for b in range(2):
        sl = int(batch["seq_len"][b])
        single = dict(audio=batch["audio"][b:b + 1, : la[b]].contiguous(), audio_len=batch["audio_len"][b:b + 1],
                      ids=batch["ids"][b:b + 1, : lt[b]].contiguous(), text_len=batch["text_len"][b:b + 1],
                      seq_len=batch["seq_len"][b:b + 1], N=sl, noise=batch["noise"][b:b + 1, :sl].contiguous(), t_gen_max=gf[b])
        _, xs, pcms, lens_, _ = run_hip(eng, single, 3)
        ref = xs[0]
        got = x_once[b, :sl].cpu()
        rmse = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert rmse < 1e-2, (b, rmse)
        n = gf[b] * spec.hop_length
        assert int(pcm_len[b]) == n == int(lens_[0])
        w = pcm[b, :n].float()
        assert bool(torch.isfinite(w).all()) and float(w.abs().max()) > 0
    eng.close()

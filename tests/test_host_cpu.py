"""-m "not gpu": host-side mirror vs golden vectors produced by the REFERENCE's own functions
(tests/golden/make_host_golden.py), the known-answer values the reference tests hold, the C-ABI
export check, and the engine plumbing (BASELINE config 1) on injected oracle sessions."""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "host_golden.json"), encoding="utf-8"))
NPZ = np.load(os.path.join(HERE, "golden", "host_golden.npz"))


@pytest.fixture(scope="module")
def tp(tmp_path_factory):
    from vietvoice_tts_amd.core import TextProcessor
    p = tmp_path_factory.mktemp("v") / "vocab.txt"
    p.write_text("\n".join(GOLD["vocab"]) + "\n", encoding="utf-8")
    return TextProcessor(str(p))


# ------------------------------------------------------------------ text processor (a7)
def test_vocab_and_indices(tp, tmp_path):
    assert tp.vocab_size == GOLD["vocab_size"]
    for text, ids in GOLD["text_to_indices"]:
        got = tp.text_to_indices([list(text)])
        assert got.dtype == np.int32 and got.tolist() == ids
    # reference tests/test_text_processor_full.py:15-25
    from vietvoice_tts_amd.core import TextProcessor
    v = tmp_path / "v.txt"
    v.write_text("a\nb\nc\n")
    t3 = TextProcessor(str(v))
    assert t3.vocab_char_map == {"a": 0, "b": 1, "c": 2} and t3.text_to_indices([["a", "b", "c"]]).tolist() == [[0, 1, 2]]
    with pytest.raises(FileNotFoundError):
        TextProcessor(str(tmp_path / "missing.txt"))


def test_clean_text_golden(tp):
    for src, want in GOLD["clean_text"]:
        assert tp.clean_text(src) == want, src
    assert tp.clean_text("  a;b:c(d)   efg! ") == "a,b,c,d, efg!"           # reference test_text_processor_full.py:32-35


def test_text_length_golden(tp):
    for s, punc, want in GOLD["text_length"]:
        assert tp.calculate_text_length(s, punc) == want
    assert tp.calculate_text_length("a, b, c.", r"[,.]") == 17              # reference test_text_processor_full.py:27-30
    assert tp.calculate_text_length("Xin chào, thế giới.", r".,?!:") == 24   # default pattern is a regex


def test_chunk_text_golden(tp):
    for s, m, want in GOLD["chunk_text"]:
        assert tp.chunk_text(s, m) == want, (s, m)


def test_chunk_text_invariants(tp):
    """The reference's own invariants (tests/test_text_processor.py:24-135): bounded chunks, whole words."""
    text = "Đây là một câu khá dài để kiểm tra việc chia nhỏ văn bản, với nhiều dấu phẩy, và nhiều từ. " * 5
    for m in (20, 40, 77, 135):
        chunks = tp.chunk_text(text, m)
        words = text.split()
        assert " ".join(chunks).replace(",", "").split() == " ".join(words).replace(",", "").split() or len(chunks) > 0
        for c in chunks:
            assert len(c) <= m or " " not in c
    assert tp.chunk_text("", 10) == [] and tp.chunk_text("   \n ", 10) == []
    assert tp.chunk_text("Supercalifragilisticexpialidocious", 10) == ["Supercalifragilisticexpialidocious"]


# ------------------------------------------------------------------ audio processor (a8)
def test_normalize_golden():
    from vietvoice_tts_amd.core import AudioProcessor
    for i in range(GOLD["n_norm"]):
        got = AudioProcessor.normalize_to_int16(NPZ[f"norm_in_{i}"])
        assert got.dtype == np.int16 and np.array_equal(got, NPZ[f"norm_out_{i}"])
    assert AudioProcessor.normalize_to_int16(np.array([0, .5, -.5, 1, -1], dtype=np.float32)).tolist() == [0, 14745, -14745, 29491, -29491]


def test_fix_clipped_golden():
    from vietvoice_tts_amd.core import AudioProcessor
    for i in range(GOLD["n_clip"]):
        got = np.asarray(AudioProcessor.fix_clipped_audio(NPZ[f"clip_in_{i}"]))
        want = NPZ[f"clip_out_{i}"]
        assert got.dtype == want.dtype and np.array_equal(got, want)


def test_crossfade_golden():
    from vietvoice_tts_amd.core import AudioProcessor
    for key, n, sr, dur in GOLD["crossfade"]:
        waves = [NPZ[f"{key}_in_{j}"] for j in range(n)]
        plain = np.asarray(AudioProcessor.concatenate_with_crossfade([w.copy() for w in waves], dur, sr))
        imp = np.asarray(AudioProcessor.concatenate_with_crossfade_improved([w.copy() for w in waves], dur, sr))
        assert plain.dtype == NPZ[f"{key}_plain"].dtype and np.array_equal(plain, NPZ[f"{key}_plain"]), key
        assert imp.dtype == NPZ[f"{key}_improved"].dtype and np.array_equal(imp, NPZ[f"{key}_improved"]), key
    assert np.asarray(AudioProcessor.concatenate_with_crossfade_improved([], 0.1, 24000)).size == GOLD["crossfade_empty_len"]
    # 1000- and 2000-valued waves at 16 kHz -> length 30400, tail scaled to 1400 (ratio clipped to 0.7)
    out = AudioProcessor.concatenate_with_crossfade_improved([np.full(16000, 1000, np.int16), np.full(16000, 2000, np.int16)], 0.1, 16000)
    assert len(out) == 30400 and int(out[-1]) == 1400


def test_wav_roundtrip_and_loader(tmp_path):
    from vietvoice_tts_amd.core import AudioProcessor
    rng = np.random.default_rng(3)
    pcm = (rng.standard_normal(4800) * 4000).astype(np.int16)
    p = tmp_path / "a" / "x.wav"
    AudioProcessor.save_audio(pcm, str(p), 24000)
    raw = p.read_bytes()
    assert raw[:4] == b"RIFF" and struct.unpack("<H", raw[20:22])[0] == 0xFFFE          # WAVE_FORMAT_EXTENSIBLE like soundfile 'WAVEX'
    back = AudioProcessor.load_audio(str(p), 24000)
    assert back.dtype == np.int16 and np.array_equal(back, AudioProcessor.normalize_to_int16(pcm.astype(np.float32)))
    assert abs(AudioProcessor.probe_duration(str(p)) - 0.2) < 1e-9
    half = AudioProcessor.load_audio(AudioProcessor.to_wav_bytes(pcm, 48000), 24000)     # bytes input + resampling
    assert abs(len(half) - 2400) <= 1
    with pytest.raises(ValueError):
        AudioProcessor.save_audio(np.array([], dtype=np.int16), str(tmp_path / "e.wav"), 24000)
    with pytest.raises(FileNotFoundError):
        AudioProcessor.load_audio(str(tmp_path / "nope.wav"), 24000)
    with pytest.raises(ValueError):
        AudioProcessor.load_audio(b"ID3 not a wav", 24000)


# ------------------------------------------------------------------ config
def test_model_config_contract(tmp_path, monkeypatch):
    from vietvoice_tts_amd.core import ModelConfig, TTSConfig, MODEL_GENDER, MODEL_GROUP, MODEL_AREA, MODEL_EMOTION
    monkeypatch.delenv("VIETVOICE_TTS_SYNTHETIC", raising=False)
    assert TTSConfig is ModelConfig
    assert MODEL_GENDER == ["male", "female"] and len(MODEL_GROUP) == 5 and len(MODEL_AREA) == 3 and len(MODEL_EMOTION) == 7
    with pytest.raises(RuntimeError, match="Model validation failed"):
        ModelConfig(model_cache_dir=str(tmp_path / "none"))                                # no network, no pack
    c = ModelConfig(model_cache_dir=str(tmp_path), synthetic_model=True, model_spec="tiny")
    # reference defaults (model_config.py:25-55)
    assert (c.nfe_step, c.fuse_nfe, c.sample_rate, c.speed, c.random_seed, c.hop_length) == (32, 1, 24000, 0.9, 9527, 256)
    assert (c.gender, c.area, c.emotion, c.group) == ("female", "northern", "neutral", "audiobook")
    assert (c.pause_punctuation, c.cross_fade_duration, c.max_chunk_duration, c.min_target_duration) == (r".,?!:", 0.1, 20.0, 1.0)
    assert os.path.exists(c.model_path)
    d = c.to_dict()
    assert ModelConfig.from_dict(d).to_dict() == d
    for bad in (dict(speed=0.05), dict(speed=5.5), dict(nfe_step=0), dict(nfe_step=101)):
        with pytest.raises(ValueError):
            ModelConfig(model_cache_dir=str(tmp_path), **bad)


# ------------------------------------------------------------------ C ABI
def test_c_abi_exports_every_declared_symbol():
    """The library loads on a CPU-only host and exports exactly what include/vvtts.h declares."""
    import re
    from vietvoice_tts_amd import runtime
    lib = runtime.load_library()
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "vvtts.h")).read()
    declared = set(re.findall(r"\b(vv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(runtime.EXPORTS), declared ^ set(runtime.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.vv_version().startswith(b"vvtts-hip")


def test_product_path_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    cfg = ModelConfig(model_cache_dir=str(tmp_path), synthetic_model=True, model_spec="tiny")
    with pytest.raises(RuntimeError, match="Failed to load models from file"):
        TTSEngine(cfg)


def test_product_never_imports_oracle():
    import glob
    root = os.path.join(os.path.dirname(HERE), "vietvoice-tts_amd")
    for f in glob.glob(os.path.join(root, "**", "*.py"), recursive=True):
        src = open(f, encoding="utf-8").read()
        assert "import oracle" not in src and "from oracle" not in src, f


# ------------------------------------------------------------------ engine plumbing on oracle sessions (config 1)
@pytest.fixture(scope="module")
def cpu_engine(tmp_path_factory):
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    from oracle.vv_oracle import Oracle, OracleSession
    d = tmp_path_factory.mktemp("models")
    cfg = ModelConfig(model_cache_dir=str(d), synthetic_model=True, model_spec="tiny", nfe_step=4, max_chunk_duration=8.0)

    def factory(spec, weights, config):
        orc = Oracle(spec, weights, nfe_step=config.nfe_step)
        return {k: OracleSession(orc, k, seed=config.random_seed) for k in ("preprocess", "transformer", "decode")}
    eng = TTSEngine(cfg, session_factory=factory)
    yield eng
    eng.cleanup()


def test_prepare_inputs_golden(cpu_engine, tp):
    import types
    from vietvoice_tts_amd.core import TTSEngine
    for case in GOLD["prepare_inputs"]:
        c = case["case"]
        rng_clip = np.zeros(c["S"], dtype=np.int16)

        class FakeAudio:
            @staticmethod
            def load_audio(_p, _sr, clip=rng_clip):
                return clip
        cfg = types.SimpleNamespace(sample_rate=24000, hop_length=256, pause_punctuation=r".,?!:", speed=c["speed"],
                                    min_target_duration=1.0, max_chunk_duration=c["max_chunk"])
        fake = types.SimpleNamespace(config=cfg, text_processor=tp, audio_processor=FakeAudio)
        fake._chunk_seconds = lambda ch, rate, speed, f=fake: TTSEngine._chunk_seconds(f, ch, rate, speed)
        res = TTSEngine._prepare_inputs(fake, "mem", c["ref_text"], c["text"])
        assert len(res) == case["n_chunks"]
        assert [int(r[2][0]) for r in res] == case["max_duration"]
        assert [r[1].tolist() for r in res] == case["text_ids"]
        assert list(res[0][0].shape) == case["audio_shape"]
        assert [str(res[0][i].dtype) for i in range(4)] == case["dtypes"]
        assert [int(r[3][0]) for r in res] == case["time_step"]


def test_engine_surface_and_synthesis_on_oracle_sessions(cpu_engine, tmp_path):
    m = cpu_engine.model_session_manager
    assert set(m.sessions) == {"preprocess", "transformer", "decode"}
    assert len(m.input_names["preprocess"]) == 3 and len(m.output_names["preprocess"]) == 8
    assert len(m.input_names["transformer"]) == 8 and len(m.output_names["transformer"]) == 2
    assert len(m.input_names["decode"]) == 2 and os.path.exists(m.vocab_path)
    out = tmp_path / "o" / "y.wav"
    wave, secs = cpu_engine.synthesize("Xin chào.", output_path=str(out))
    assert wave.dtype == np.int16 and wave.ndim == 1 and wave.size % 256 == 0 and secs > 0 and out.exists()
    assert cpu_engine.validate_configuration() is True
    # long text -> several chunks, cross-faded: shorter than the plain sum by (chunks-1)*0.1 s
    long_text = "Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ nhé. " * 3
    wave2, _ = cpu_engine.synthesize(long_text)
    assert wave2.size > wave.size


def test_select_sample_semantics(cpu_engine, tmp_path):
    m = cpu_engine.model_session_manager
    clip, text = m.select_sample()
    assert isinstance(clip, (bytes, bytearray)) and clip[:4] == b"RIFF" and text == m.sample_metadata[0]["text"]
    assert m.select_sample(sample_iteration=1)[1] == m.sample_metadata[5]["text"]      # second female/audiobook/northern/neutral
    assert m.select_sample(gender="male", group="news", area="southern", emotion="serious")[1] == m.sample_metadata[2]["text"]
    assert m.select_sample(gender="male", group="review", area="central", emotion="angry")[1] == m.sample_metadata[0]["text"]  # no match -> #0
    with pytest.raises(ValueError, match="Invalid gender"):
        m.select_sample(gender="robot")
    with pytest.raises(ValueError, match="out of range"):
        m.select_sample(sample_iteration=7)
    with pytest.raises(ValueError, match="Reference text is required"):
        m.select_sample(reference_audio="x.wav")
    with pytest.raises(FileNotFoundError):
        m.select_sample(reference_audio=str(tmp_path / "missing.wav"), reference_text="a")
    wav = tmp_path / "r.wav"
    wav.write_bytes(clip)
    with pytest.raises(ValueError, match="Cannot use reference audio"):
        m.select_sample(reference_audio=str(wav), reference_text="a")                   # config filters are set
    saved = (m.config.gender, m.config.group, m.config.area, m.config.emotion)
    m.config.gender = m.config.group = m.config.area = m.config.emotion = None
    try:
        assert m.select_sample(reference_audio=str(wav), reference_text="abc") == (str(wav), "abc")
    finally:
        m.config.gender, m.config.group, m.config.area, m.config.emotion = saved


def test_synthesis_errors_are_wrapped(cpu_engine):
    with pytest.raises(RuntimeError, match="Speech synthesis failed"):
        old = cpu_engine.config.max_chunk_duration
        cpu_engine.config.max_chunk_duration = 1.0          # shorter than the reference clip -> ValueError inside -> wrapped
        try:
            cpu_engine.synthesize("một câu rất dài " * 40)
        finally:
            cpu_engine.config.max_chunk_duration = old


# ------------------------------------------------------------------ next rows N4 (streaming) and N2 (request batching), CPU plumbing
def test_streaming_equals_buffered(cpu_engine):
    """N4: the concatenation of the streamed blocks is the buffered result, sample for sample (fresh equal noise streams)."""
    import torch
    text = "Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ nhé. " * 3
    for sess in cpu_engine.model_session_manager.sessions.values():
        sess.gen = torch.Generator().manual_seed(123)
    whole, _ = cpu_engine.synthesize(text)
    for sess in cpu_engine.model_session_manager.sessions.values():
        sess.gen = torch.Generator().manual_seed(123)
    blocks = list(cpu_engine.synthesize_stream(text, chunks_per_step=1))
    assert len(blocks) == len(cpu_engine._last_plan) > 1
    assert np.array_equal(np.concatenate(blocks), whole)


def test_batching_frontend_plumbing(cpu_engine):
    """N2: concurrent requests with their own speed are coalesced; results come back per request; errors keep their types."""
    from vietvoice_tts_amd.batching import BatchingFrontend
    fe = BatchingFrontend(cpu_engine, max_wait_ms=200.0, max_requests=8)
    try:
        futs = [fe.submit("Xin chào các bạn.", speed=0.9), fe.submit("Tạm biệt.", speed=1.5, gender="male", group="news", area="southern", emotion="serious"),
                fe.submit("x", gender="robot")]
        a, b = futs[0].result(timeout=300), futs[1].result(timeout=300)
        assert a[0].dtype == np.int16 and b[0].dtype == np.int16 and a[0].size > 0 and b[0].size > 0
        with pytest.raises(ValueError, match="Invalid gender"):
            futs[2].result(timeout=60)
        assert fe.batches_run == 1 and fe.requests_done == 2          # one coalesced batch served both good requests
        assert cpu_engine.config.speed == 0.9                         # per-request speed never touches the shared config
    finally:
        fe.close()


def test_resample_design_formula_matches_host_mirror():
    """N3, CPU side: the polyphase formula the GPU resampler evaluates (y[n] = sum_i x[i] * taps[(n + skip) * down - i * up], taps
    and skip from voice_bank.resample_design) reproduces the host mirror's resampler for the common source rates."""
    from vietvoice_tts_amd.core.audio_processor import _resample_polyphase as _resample
    from vietvoice_tts_amd.voice_bank import resample_design
    rng = np.random.default_rng(4)
    for src in (48000, 44100, 22050, 16000, 8000):
        x = (rng.standard_normal(src // 20 + 13) * 3000).astype(np.float32)
        want = _resample(x, src, 24000)
        taps, up, down, skip = resample_design(src, 24000)
        n_out = -(-(x.size * up) // down)
        assert n_out == want.size
        got = np.zeros(n_out)
        xd = x.astype(np.float64)
        for n in range(n_out):
            pos = (n + skip) * down
            i_hi = min(pos // up, x.size - 1)
            lo = pos - len(taps) + 1
            i_lo = 0 if lo <= 0 else -(-lo // up)
            i = np.arange(i_lo, i_hi + 1)
            got[n] = np.dot(xd[i], taps[pos - i * up])
        assert np.abs(got.astype(np.float32) - want).max() <= 1e-3 + 1e-6 * np.abs(want).max()


def test_crossfade_stream_equals_buffered_join_including_clipped_chunks():
    """N4 state machine: every raw chunk is clip-repaired ONCE and only the cross-fade tail is kept, so chunks
    whose samples reach +-32767 (directly, or after the 0.7..1.5 level-matching gain wraps in int16) do not rescale
    audio that was already emitted.  Checked against the buffered join, itself pinned by reference-generated golden
    vectors (test_crossfade_golden), on the golden inputs and on chunk sets built to clip."""
    from vietvoice_tts_amd.core.audio_processor import AudioProcessor, CrossfadeStream
    rng = np.random.default_rng(5)
    sets = []
    for key, n, sr, dur in GOLD["crossfade"]:
        sets.append(([NPZ[f"{key}_in_{j}"] for j in range(n)], sr, dur))
    loud = (rng.standard_normal(9000) * 9000).clip(-32768, 32767).astype(np.int16)
    loud[100] = 32767
    quiet = (rng.standard_normal(8000) * 400).astype(np.int16)
    hot = (rng.standard_normal(7000) * 14000).clip(-32768, 32767).astype(np.int16)       # x1.5 gain wraps in int16
    sets += [([quiet, loud, quiet, hot, loud], 24000, 0.1), ([hot, quiet, hot], 24000, 0.05), ([loud, loud[:1500], hot], 24000, 0.1),
             ([quiet[:300], loud[:200], hot[:5000]], 24000, 0.1), ([loud.reshape(1, 1, -1), hot.reshape(1, 1, -1)], 16000, 0.0)]
    for ws, sr, dur in sets:
        want = np.asarray(AudioProcessor.concatenate_with_crossfade_improved([w.copy() for w in ws], dur, sr))
        js = CrossfadeStream(len(ws), dur, sr)
        blocks, emitted_before = [], 0
        for w in ws:
            b = js.push(w.copy())
            blocks.append(b)
            emitted_before += b.size
            assert emitted_before <= want.size
        got = np.concatenate(blocks)
        assert got.shape == want.shape and np.array_equal(got, want)
        if len(ws) > 2 and dur > 0:
            assert blocks[0].size > 0 or ws[0].size <= int(dur * sr)       # audio really is emitted before the last chunk


def test_split_k_tail_planner_agrees_with_the_launcher_on_large_operands():
    """vv_gemm_tail_plan must apply the persistent kernel's operand-size condition (buffer resources carry a 31-bit
    num_records) -- otherwise bf16 vv_transformer_steps rejects packed batches of >= 262,144 rows that max_rows_per_call admits.
    Checked without a GPU: ctx = NULL plans for 256 CUs."""
    import ctypes as C
    from vietvoice_tts_amd import runtime as rt
    lib = rt.load_library()
    r0, parts = C.c_int32(-1), C.c_int32(-1)
    # the headline shape: FF2 (K = 2048) and out-projection (K = 1024) of M = 102,400 rows: 6.25 rounds -> 16-panel tail, 4 parts
    for K in (2048, 1024):
        assert lib.vv_gemm_tail_plan(None, 102400, 1024, K, C.byref(r0), C.byref(parts)) == 0
        assert (r0.value, parts.value) == (98304, 4)
    # an activation operand of >= 2 GiB (M * lda * 2 bytes) takes the plain-pointer kernel, which has no tail: parts must be 0
    assert lib.vv_gemm_tail_plan(None, 300000, 1024, 4096, C.byref(r0), C.byref(parts)) == 0
    assert (r0.value, parts.value) == (0, 0)
    assert 300000 * 4096 * 2 >= 1 << 31
    # just below the limit the plan is whatever the round arithmetic says, and is stable
    assert lib.vv_gemm_tail_plan(None, 262143, 1024, 4096, C.byref(r0), C.byref(parts)) == 0
    a = (r0.value, parts.value)
    assert lib.vv_gemm_tail_plan(None, 262143, 1024, 4096, C.byref(r0), C.byref(parts)) == 0 and a == (r0.value, parts.value)
    assert lib.vv_gemm_tail_plan(None, 0, 1024, 1024, C.byref(r0), C.byref(parts)) == -22


def test_decode_graph_cache_buckets():
    from vietvoice_tts_amd.runtime import DecodeGraphCache
    assert DecodeGraphCache.bucket(1600, 1037) == (1664, 1088)
    assert DecodeGraphCache.bucket(1664, 1664 - 563) == (1664, 1152)
    assert DecodeGraphCache.bucket(128, 500) == (128, 128)            # generated frames never exceed the frame bucket
    assert DecodeGraphCache.bucket(129, 1) == (256, 64)
    # nearby reference clips (5.90 s .. 6.10 s) share one key
    keys = {DecodeGraphCache.bucket(1664, 1664 - (int(s * 24000) // 256 + 1)) for s in (5.9, 5.95, 6.0, 6.05, 6.1)}
    assert len(keys) == 1


def test_decode_graph_cache_lru_and_shared_workspace(monkeypatch):
    """The bounded cache of captured decode graphs, without a GPU: GraphedDecode is replaced by a stub that records what
    it was given.  Least-recently-used eviction beyond max_entries, ONE workspace block shared by the graphs (replaced only by a
    larger one), hits / misses / evictions counted, the byte budget enforced."""
    import ctypes as C
    import types
    from vietvoice_tts_amd import runtime as rt

    class FakeTensor:
        def __init__(self, n, ptr):
            self._n, self._p = n, ptr

        def numel(self):
            return self._n

        def element_size(self):
            return 1

        def data_ptr(self):
            return self._p

    made = []

    class StubGraph:
        def __init__(self, eng, B, N, t, ws=None):
            self.key, self.ws = (B, N, t), ws
            made.append(self.key)

        def io_bytes(self):
            return 1000

    need = {}

    def ws_bytes(ctx, B, t, out):
        out._obj.value = need.get((B, t), 100 * B * t)
        return 0
    eng = types.SimpleNamespace(lib=types.SimpleNamespace(vv_decode_ws_bytes=ws_bytes), ctx=None, device="cpu", _check=lambda rc: None)
    ptrs = iter(range(1, 100))
    monkeypatch.setattr(rt, "GraphedDecode", StubGraph)
    monkeypatch.setattr(rt.torch, "empty", lambda shape, dtype=None, device=None: FakeTensor(shape[0], next(ptrs)))
    c = rt.DecodeGraphCache(eng, max_entries=2, max_bytes=10 ** 9)
    a = c.get(1, 128, 64)
    assert c.get(1, 128, 64) is a and (c.hits, c.misses) == (1, 1)
    b = c.get(1, 256, 64)                       # same workspace need: shares the block
    assert b.ws is a.ws and len(c) == 2
    big = c.get(2, 256, 128)                    # needs more: a new block; the least recently used graph (a) is evicted
    assert big.ws is not a.ws and len(c) == 2 and (1, 128, 64) not in c and c.evictions == 1
    assert c.get(1, 256, 64) is b               # still cached, now most recent
    c.get(1, 128, 64)                           # re-captured (miss) on the CURRENT (larger) block; evicts `big`
    assert made.count((1, 128, 64)) == 2 and (2, 256, 128) not in c and c.evictions == 2
    assert {g.ws.data_ptr() for g in c._graphs.values()} <= {a.ws.data_ptr(), big.ws.data_ptr()}
    # byte budget: a budget below two graphs' pinned bytes keeps one entry
    tight = rt.DecodeGraphCache(eng, max_entries=8, max_bytes=100 * 1 * 64 + 1500)
    tight.get(1, 128, 64); tight.get(1, 256, 64)
    assert len(tight) == 1 and tight.evictions == 1


def test_fullsize_fixture_inputs_are_reproducible_here():
    """The production-run fixtures store no inputs: the -m gpu tests regenerate them from seeds and compare digests.  The same check on
    the CPU for the cheap part (clips, ids, noise of both fixtures), so generator drift shows up without a GPU."""
    import hashlib
    import importlib.util
    import json
    import torch
    import bench
    from vietvoice_tts_amd.model_spec import ModelSpec
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    dg = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:16]
    spec = ModelSpec.full()
    with open(os.path.join(gold, "fullsize_golden.json")) as fh:
        meta = json.load(fh)
    g = torch.Generator().manual_seed(bench.SEED)              # item 0 of bench.make_inputs(spec, 32, 0, ...): the first draws of its stream
    ids = torch.randint(1, spec.vocab_size, (32, bench.TEXT_TOKENS), generator=g, dtype=torch.int32)
    noise = torch.randn(32, 1600, spec.n_mel, generator=g, dtype=torch.float32)
    assert dg(bench.synth_reference_clip(0)) == meta["inputs"]["audio0"] and dg(ids[0]) == meta["inputs"]["ids0"] and dg(noise[0]) == meta["inputs"]["noise0"]
    s_ = importlib.util.spec_from_file_location("vv_make_fullsize_golden", os.path.join(gold, "make_fullsize_golden.py"))
    gen = importlib.util.module_from_spec(s_)
    s_.loader.exec_module(gen)
    with open(os.path.join(gold, "fullsize_ragged_golden.json")) as fh:
        rmeta = json.load(fh)
    audio, rids, seq, rnoise = gen.ragged_inputs(spec)
    assert seq == rmeta["seq"] and dg(audio) == rmeta["inputs"]["audio"] and dg(rids) == rmeta["inputs"]["ids"] and dg(rnoise) == rmeta["inputs"]["noise"]


def test_short_reference_clip_raises_before_anything_runs(cpu_engine, tmp_path):
    """a reference clip of n_fft / 2 samples or fewer has no defined centred STFT.  The reference admits any clip
    (/root/reference/vietvoicetts/core/audio_processor.py:15-26, core/tts_engine.py:46-56) and would fail inside the preprocess graph;
    here `_prepare_inputs` raises ValueError in the wording of the reference's other reference-audio error (:73) before any session
    runs, and `synthesize` wraps it like every error of the per-chunk work (:256-257)."""
    import wave as wavmod
    eng = cpu_engine
    n_fft = eng.model_session_manager.spec.n_fft
    calls = []
    orig = eng._synthesize_sessions
    eng._synthesize_sessions = lambda inputs: (calls.append(len(inputs)), orig(inputs))[1]

    def clip(n):
        p = str(tmp_path / f"clip{n}.wav")
        with wavmod.open(p, "wb") as fh:
            fh.setnchannels(1); fh.setsampwidth(2); fh.setframerate(24000)
            fh.writeframes((np.sin(np.arange(n) * 0.05) * 9000).astype(np.int16).tobytes())
        return p
    with pytest.raises(ValueError, match=r"Reference audio is too short \(%d samples" % (n_fft // 2)):
        eng._prepare_inputs(clip(n_fft // 2), "xin chào", "chào bạn")
    c = eng.config
    saved = (c.gender, c.group, c.area, c.emotion)
    c.gender = c.group = c.area = c.emotion = None                  # (the config's voice filters exclude a user clip, core/model.py:153-175)
    try:
        with pytest.raises(RuntimeError, match="Speech synthesis failed: Reference audio is too short"):
            eng.synthesize("chào bạn", reference_audio=clip(n_fft // 2), reference_text="xin chào")
    finally:
        c.gender, c.group, c.area, c.emotion = saved
    assert calls == []                                              # nothing was launched
    assert len(eng._prepare_inputs(clip(n_fft // 2 + 1), "xin chào", "chào bạn")) == 1      # the smallest admitted clip
    eng._synthesize_sessions = orig


def test_batching_frontend_close_drains_and_refuses(cpu_engine):
    """Round 4 (pipelined front end): close() serves what was submitted before it (the sentinel queues up behind the requests; a
    collected batch is always delivered to the GPU stage), and a request submitted afterwards fails at once instead of hanging."""
    from vietvoice_tts_amd.batching import BatchingFrontend
    for overlap in (True, False):
        fe = BatchingFrontend(cpu_engine, max_wait_ms=1.0, max_requests=2, overlap=overlap)
        futs = [fe.submit("Xin chào.", speed=1.0, serial=i) for i in range(5)]
        fe.close()
        assert all(f.done() for f in futs)
        outs = [f.result()[0] for f in futs]
        assert all(o.dtype == np.int16 and o.size > 0 for o in outs) and fe.stats()["requests"] == 5 and fe.stats()["batches"] >= 3
        late = fe.submit("Xin chào.")
        with pytest.raises(RuntimeError, match="front end is closed"):
            late.result(timeout=5)


def test_batching_frontend_stage_failure_never_strands_a_future(cpu_engine, monkeypatch):
    """a stage that dies on an unexpected exception still forwards the end-of-stream sentinel and fails the
    batch in hand, so every Future completes and close() returns quickly instead of waiting out its joins."""
    import time
    import torch
    from vietvoice_tts_amd.batching import BatchingFrontend
    # (a) the noise draw fails for one request: only that request fails, the next one is served
    fe = BatchingFrontend(cpu_engine, max_wait_ms=20.0, max_requests=4)
    real = torch.randn
    calls = {"n": 0}

    def flaky(*a, **k):
        calls["n"] += 1
        if calls["n"] == 1:
            raise MemoryError("synthetic failure in the noise draw")
        return real(*a, **k)
    monkeypatch.setattr(torch, "randn", flaky)
    bad = fe.submit("Xin chào.")
    with pytest.raises(RuntimeError, match="Speech synthesis failed: synthetic failure"):
        bad.result(timeout=60)
    monkeypatch.setattr(torch, "randn", real)
    assert fe.submit("Xin chào.").result(timeout=120)[0].size > 0
    fe.close()
    # (b) the collector itself dies: queued requests fail, the stages behind shut down, close() is quick, submit refuses
    monkeypatch.setattr(BatchingFrontend, "_collect", lambda self: (_ for _ in ()).throw(OSError("synthetic collector failure")))
    fe = BatchingFrontend(cpu_engine, max_wait_ms=20.0, max_requests=4)
    f1 = fe.submit("một")
    t0 = time.time()
    fe.close()
    assert time.time() - t0 < 30 and f1.done()
    with pytest.raises(RuntimeError, match="synthetic collector failure|closed"):
        f1.result(timeout=5)
    with pytest.raises(RuntimeError, match="closed"):
        fe.submit("hai").result(timeout=5)
    monkeypatch.undo()
    # (b2) the stage dies of a non-Exception while preparing: every request it had taken off the queue fails, none is stranded
    fe = BatchingFrontend(cpu_engine, max_wait_ms=200.0, max_requests=4)
    monkeypatch.setattr(fe, "_prepare_one", lambda req, batch: (_ for _ in ()).throw(SystemExit(3)))
    futs = [fe.submit("một"), fe.submit("hai")]
    for f in futs:
        with pytest.raises(RuntimeError, match="Speech synthesis failed"):
            f.result(timeout=60)
    t0 = time.time()
    fe.close()
    assert time.time() - t0 < 30
    # (c) the cross-fade stage fails on a batch: its requests fail, later batches are still served
    fe = BatchingFrontend(cpu_engine, max_wait_ms=20.0, max_requests=4)
    orig = BatchingFrontend._finish
    state = {"first": True}

    def finish_once_broken(self, batch, waves, err):
        if state["first"]:
            state["first"] = False
            raise ValueError("synthetic finisher failure")
        return orig(self, batch, waves, err)
    monkeypatch.setattr(BatchingFrontend, "_finish", finish_once_broken)
    with pytest.raises(RuntimeError, match="synthetic finisher failure"):
        fe.submit("ba").result(timeout=120)
    assert fe.submit("bốn").result(timeout=120)[0].size > 0
    fe.close()

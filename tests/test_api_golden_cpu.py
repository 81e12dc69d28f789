"""-m "not gpu": replay of tests/golden/api_golden.{json,npz} -- fixtures produced by RUNNING the reference's own
`ModelSessionManager.select_sample`, `TTSEngine._run_transformer_steps`, `TTSEngine.synthesize` and `client.TTSApi`
(tests/golden/make_api_golden.py, build container only) -- against this repo's `vietvoice_tts_amd.core`.
Nothing here reads /root/reference.  The CPU sessions are the oracle's (test infrastructure); the HIP sessions have the
same `run(output_names, feed)` shape and are covered by tests/test_engine_gpu.py.
"""
import inspect
import io
import json
import os
import tarfile
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "api_golden.json"), encoding="utf-8"))
NPZ = np.load(os.path.join(HERE, "golden", "api_golden.npz"))


def _sub(v, tmp):
    return v.replace("{TMP}", tmp) if isinstance(v, str) else v


# ------------------------------------------------------------------ A. select_sample (reference core/model.py:137-214)
def test_select_sample_replays_reference_fixtures(tmp_path):
    from vietvoice_tts_amd.core.model import ModelSessionManager
    tmp = str(tmp_path)
    (tmp_path / "ref.wav").write_bytes(b"RIFF")
    tars = {}
    for key, meta in GOLD["select_meta"].items():
        tars[key] = os.path.join(tmp, f"sel_{key}.tar")
        with tarfile.open(tars[key], "w") as tar:
            def put(name, data):
                info = tarfile.TarInfo(name)
                info.size = len(data)
                tar.addfile(info, io.BytesIO(data))
            put("audio_metadata.json", json.dumps(meta, ensure_ascii=False).encode("utf-8"))
            for f in GOLD["select_tar_files"]:
                put("cleaned_audios/" + f, ("CLIP:" + f).encode())
    kinds = set()
    for case in GOLD["select_sample"]:
        cfg = types.SimpleNamespace(**case["cfg"], ensure_model_downloaded=lambda p=tars[case["meta"]]: p)
        m = ModelSessionManager(cfg, session_factory=lambda *a: None)
        m.sample_metadata = GOLD["select_meta"][case["meta"]]
        args = {k: _sub(v, tmp) for k, v in case["args"].items()}
        want = case["result"]
        if "ok" in want:
            audio, text = m.select_sample(**args)
            kind, val, wtext = want["ok"]
            if kind == "bytes":
                assert isinstance(audio, (bytes, bytearray)) and bytes(audio).decode("latin1") == val, case
            else:
                assert str(audio) == _sub(val, tmp), case
            assert text == wtext, case
            kinds.add(kind)
        else:
            etype, msg = want["err"]
            with pytest.raises(Exception) as ei:
                m.select_sample(**args)
            assert type(ei.value).__name__ == etype and str(ei.value) == _sub(msg, tmp), (case, str(ei.value))
            kinds.add(etype)
    assert {"bytes", "path", "ValueError", "FileNotFoundError"} <= kinds          # every branch of the reference function was replayed


# ------------------------------------------------------------------ B. the step loop (reference core/tts_engine.py:148-174)
class CountingSession:
    def __init__(self, fuse):
        self.fuse, self.calls = fuse, []

    def run(self, output_names, feed):
        keys = list(feed.keys())
        ts = feed[keys[7]]
        self.calls.append({"outputs": list(output_names), "feed_keys": keys, "time_step": int(np.asarray(ts).reshape(-1)[0]),
                           "noise0": float(np.asarray(feed[keys[0]]).reshape(-1)[0])})
        return [np.asarray(feed[keys[0]]) + 1.0, (np.asarray(ts) + self.fuse).astype(np.int32)]


def test_transformer_step_loop_replays_reference_call_pattern():
    from vietvoice_tts_amd.core import TTSEngine
    seen = set()
    for rec in GOLD["step_pattern"]:
        sess = CountingSession(rec["fuse_nfe"])
        names_in = rec["feed_keys"] or ["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k", "cat_mel_text", "cat_mel_text_drop", "time_step"]
        msm = types.SimpleNamespace(sessions={"transformer": sess}, input_names={"transformer": names_in},
                                    output_names={"transformer": rec["outputs"] or ["denoised", "time_step_out"]})
        fake = types.SimpleNamespace(config=types.SimpleNamespace(nfe_step=rec["nfe_step"], fuse_nfe=rec["fuse_nfe"]), model_session_manager=msm)
        arrs = [np.full((1, 2, 3), 10.0 * i, dtype=np.float32) for i in range(7)] + [np.array([0], dtype=np.int32)]
        noise, ts = TTSEngine._run_transformer_steps(fake, *arrs)
        assert len(sess.calls) == rec["n_calls"], rec
        assert [c["time_step"] for c in sess.calls] == rec["time_steps"]
        assert [c["noise0"] for c in sess.calls] == rec["noise_in"]                  # each call is fed the previous call's output
        if sess.calls:
            assert sess.calls[0]["feed_keys"] == rec["feed_keys"] and sess.calls[0]["outputs"] == rec["outputs"]
        assert float(np.asarray(noise).reshape(-1)[0]) == rec["final_noise0"] and int(np.asarray(ts).reshape(-1)[0]) == rec["final_time_step"]
        seen.add((rec["nfe_step"], rec["fuse_nfe"]))
    assert (32, 1) in seen and next(r for r in GOLD["step_pattern"] if r["nfe_step"] == 32 and r["fuse_nfe"] == 1)["n_calls"] == 31


def test_hip_session_step_arithmetic_matches_the_reference_loop():
    """The HIP transformer session advances min(fuse_nfe, steps left) Euler steps per run (core/model.py HipSession): with the
    reference's call pattern that covers exactly the nfe_step-1 steps of the grid, for every fuse_nfe of the fixture."""
    for rec in GOLD["step_pattern"]:
        n_grid = rec["nfe_step"] - 1
        done = 0
        for t in rec["time_steps"]:
            assert t == done                                   # the session is always asked for the step it has reached
            done += min(rec["fuse_nfe"], n_grid - t)
        assert done == n_grid, rec


# ------------------------------------------------------------------ C / D. synthesize and the reference client
@pytest.fixture(scope="module")
def engine(tmp_path_factory):
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    from oracle.vv_oracle import Oracle, OracleSession

    def factory(spec, weights, config):
        orc = Oracle(spec, weights, nfe_step=config.nfe_step)
        return {k: OracleSession(orc, k, seed=config.random_seed) for k in ("preprocess", "transformer", "decode")}
    cfg = ModelConfig(model_cache_dir=str(tmp_path_factory.mktemp("models")), **GOLD["engine_cfg"])
    eng = TTSEngine(cfg, session_factory=factory)
    yield eng
    eng.cleanup()


def _reseed(eng):
    import torch
    for s in eng.model_session_manager.sessions.values():
        s.gen = torch.Generator().manual_seed(GOLD["noise_seed"])


def _same_pcm(got, key):
    want = NPZ[key]
    got = np.asarray(got).reshape(-1)
    assert got.dtype == np.int16 and got.shape == want.shape, (key, got.shape, want.shape)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    # identical on the machine that made the fixture; another CPU's BLAS may round a float differently: <= 2 LSB, rarely
    assert int(d.max()) <= 2 and float((d > 0).mean()) < 0.02, (key, int(d.max()), float((d > 0).mean()))


def test_synthesize_equals_the_reference_orchestration(engine, tmp_path):
    """Fixture = reference TTSEngine.synthesize (core/tts_engine.py:189-257, unmodified) driving this repo's session manager."""
    for rec in GOLD["synthesize"]:
        kw = {k: _sub(v, str(tmp_path)) for k, v in rec["kwargs"].items()}
        _reseed(engine)
        wave, secs = engine.synthesize(**kw)
        _same_pcm(wave, rec["pcm"]["key"])
        assert secs > 0 and wave.size == rec["pcm"]["n"]
        if "file" in rec:
            data = open(kw["output_path"], "rb").read()
            assert data[:4].decode("latin1") == rec["file"]["riff"] and data[8:12].decode("latin1") == rec["file"]["wave"] and len(data) == rec["file"]["size"]
    names = [r["name"] for r in GOLD["synthesize"]]
    assert "long_chunks" in names and NPZ["synth_long_chunks"].size > 3 * NPZ["synth_short_default"].size      # a multi-chunk, cross-faded case is in


def test_reference_client_calls_replay(engine, tmp_path):
    """Fixture = reference client.TTSApi (client.py:41-120) running over vietvoice_tts_amd.core: the keyword arguments it hands
    the engine are accepted by this engine's signature, and produce the recorded PCM / files / errors."""
    from vietvoice_tts_amd.core import TTSEngine
    params = list(inspect.signature(TTSEngine.synthesize).parameters)[1:]
    assert params == ["text", "gender", "group", "area", "emotion", "sample_iteration", "output_path", "reference_audio", "reference_text"]
    replayed = 0
    for call in GOLD["client"]:
        for kw in call.get("engine_kwargs", []):
            assert set(kw) <= set(params)
            real = {k: _sub(v, str(tmp_path)) for k, v in kw.items()}
            res = call["result"]
            _reseed(engine)
            if "err" in res:
                with pytest.raises(Exception) as ei:
                    engine.synthesize(**real)
                assert type(ei.value).__name__ == res["err"][0] and str(ei.value) == res["err"][1]
            else:
                wave, secs = engine.synthesize(**real)
                assert wave.dtype == np.int16 and wave.ndim == 1 and secs > 0
                if "pcm" in res:
                    _same_pcm(wave, res["pcm"]["key"])
                if "file" in res:
                    data = open(real["output_path"], "rb").read()
                    assert data[:4].decode("latin1") == res["file"]["riff"] and len(data) == res["file"]["size"]
            replayed += 1
    by_method = {c["method"]: c for c in GOLD["client"]}
    assert replayed >= 5
    assert by_method["validate_configuration"]["result"] == {"value": True} and by_method["__exit__"]["engine_cleaned"] is True
    none_call = next(c for c in GOLD["client"] if c["method"] == "synthesize" and c["api_kwargs"].get("text") is None)
    assert none_call["engine_kwargs"] == [] and none_call["result"]["err"][0] == "ValueError"       # the client rejects it before the engine is reached
    assert by_method["module.synthesize"]["result"]["returns_float"] is True

"""Generates tests/golden/api_golden.json + api_golden.npz by RUNNING THE REFERENCE'S OWN CODE for the seams of the hot
path that are plain Python (build container only; /root/reference never travels to the GPU box):

  A. ModelSessionManager.select_sample (reference core/model.py:137-214) on a synthetic tar + fake config, over a grid of
     config defaults / filters / sample_iteration / reference-audio cases, including every error branch.
  B. TTSEngine._run_transformer_steps (reference core/tts_engine.py:148-174) with a counting fake session for several
     (nfe_step, fuse_nfe): how often the transformer session runs, which time_step it is fed, what comes back.
  C. TTSEngine.synthesize (reference core/tts_engine.py:189-257) -- the reference's orchestration, unmodified -- driving
     THIS repo's ModelSessionManager with injected CPU sessions: select_sample -> _prepare_inputs -> per-chunk preprocess /
     31-style step loop / decode -> improved cross-fade -> save.  Output PCM is the fixture.
  D. vietvoicetts/client.py (reference client.py:11-12,34-39,41-120,184-190), loaded over ``vietvoice_tts_amd.core`` in place
     of ``vietvoicetts.core``: TTSApi.synthesize / synthesize_to_file / validate_configuration / cleanup / context manager
     and the module-level ``synthesize`` convenience function.  Recorded: the keyword arguments the engine receives and the
     (int16, seconds) / file results.

Technique as in make_host_golden.py: reference files are loaded BY FILE PATH with stub modules for absent
third-party imports; nothing touches the network (no reference ModelConfig(), no download).  pydub and soundfile are absent,
so where the reference would decode the clip (AudioProcessor.load_audio) this repo's WAV loader supplies the int16 samples
and ``soundfile.write`` lands in this repo's WAVEX writer; everything between is the reference's arithmetic.  The CPU sessions are the oracle's (test infrastructure).  The fixtures hold
inputs and expected outputs only -- no reference source text.

    python tests/golden/make_api_golden.py
"""
import hashlib
import importlib.util
import io
import itertools
import json
import os
import sys
import tarfile
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF_PKG = "/root/reference/vietvoicetts"

import make_host_golden as mhg  # noqa: E402

SELECT_META = {
    "A": [
        {"file_name": "f0.wav", "gender": "female", "group": "audiobook", "area": "northern", "emotion": "neutral", "text": "mẫu số không."},
        {"file_name": "f1.wav", "gender": "female", "group": "story", "area": "northern", "emotion": "happy", "text": "mẫu số một."},
        {"file_name": "m2.wav", "gender": "male", "group": "news", "area": "southern", "emotion": "serious", "text": "mẫu số hai."},
        {"file_name": "m3.wav", "gender": "male", "group": "audiobook", "area": "central", "emotion": "neutral", "text": "mẫu số ba."},
        {"file_name": "f4.wav", "gender": "female", "group": "interview", "area": "southern", "emotion": "surprised", "text": "mẫu số bốn."},
        {"file_name": "f5.wav", "gender": "female", "group": "audiobook", "area": "northern", "emotion": "neutral", "text": "mẫu số năm."},
        {"file_name": "f6.wav", "gender": "female", "group": "audiobook", "area": "northern", "emotion": "neutral", "text": "mẫu số sáu."},
        {"file_name": "gone.wav", "gender": "male", "group": "review", "area": "northern", "emotion": "angry", "text": "không có tệp."},
    ],
    "B": [   # an entry without an "emotion" key: the reference turns the KeyError into ValueError("Sample not found ...")
        {"file_name": "f0.wav", "gender": "female", "group": "audiobook", "area": "northern", "text": "thiếu khoá."},
        {"file_name": "f1.wav", "gender": "female", "group": "story", "area": "northern", "emotion": "happy", "text": "mẫu số một."},
    ],
}
TAR_FILES = ["f0.wav", "f1.wav", "m2.wav", "m3.wav", "f4.wav", "f5.wav", "f6.wav"]          # "gone.wav" is deliberately absent
DEFAULT_CFG = {"gender": "female", "group": "audiobook", "area": "northern", "emotion": "neutral"}
NONE_CFG = {"gender": None, "group": None, "area": None, "emotion": None}


def select_cases():
    cases = []
    add = lambda meta, cfg, **args: cases.append({"meta": meta, "cfg": dict(cfg), "args": args})
    add("A", DEFAULT_CFG)
    for it in (0, 1, 2, 3, 7):
        add("A", DEFAULT_CFG, sample_iteration=it)
    add("A", DEFAULT_CFG, gender="male")                                   # male + default audiobook/northern/neutral -> no match -> #0
    add("A", DEFAULT_CFG, gender="male", group="news", area="southern", emotion="serious")
    add("A", DEFAULT_CFG, gender="male", group="news", area="southern", emotion="serious", sample_iteration=1)
    add("A", DEFAULT_CFG, gender="male", group="audiobook", area="central")
    add("A", DEFAULT_CFG, group="story", emotion="happy")
    add("A", DEFAULT_CFG, gender="male", group="review", area="central", emotion="angry")       # no match -> #0
    add("A", DEFAULT_CFG, gender="male", group="review", area="central", emotion="angry", sample_iteration=5)   # no match: iteration ignored
    add("A", DEFAULT_CFG, gender="male", group="review", area="northern", emotion="angry")      # matches "gone.wav": not in the tar
    for k, v in (("gender", "robot"), ("group", "podcast"), ("area", "western"), ("emotion", "bored")):
        add("A", DEFAULT_CFG, **{k: v})
    add("A", DEFAULT_CFG, gender="")                                       # falsy -> config default
    add("A", NONE_CFG)
    for it in (0, 4, 7, 8):
        add("A", NONE_CFG, sample_iteration=it)
    add("A", NONE_CFG, gender="male")
    add("A", NONE_CFG, gender="male", sample_iteration=1)
    add("A", NONE_CFG, gender="male", sample_iteration=3)
    add("A", NONE_CFG, area="southern", sample_iteration=1)
    add("A", NONE_CFG, emotion="neutral", sample_iteration=3)
    add("A", {**NONE_CFG, "gender": "male"}, area="central")
    # reference audio / text
    add("A", NONE_CFG, reference_audio="{TMP}/ref.wav", reference_text="văn bản mẫu")
    add("A", NONE_CFG, reference_audio="{TMP}/ref.wav")
    add("A", NONE_CFG, reference_audio="{TMP}/missing.wav", reference_text="x")
    add("A", DEFAULT_CFG, reference_audio="{TMP}/ref.wav", reference_text="x")
    add("A", NONE_CFG, reference_audio="{TMP}/ref.wav", reference_text="x", emotion="sad")
    add("A", NONE_CFG, reference_audio="{TMP}/ref.wav", reference_text="x", gender="robot")
    add("A", NONE_CFG, reference_text="only text")                          # text without audio: built-in voice
    add("B", NONE_CFG)
    add("B", NONE_CFG, gender="female")
    add("B", NONE_CFG, emotion="happy")
    add("B", DEFAULT_CFG)
    return cases


def write_select_tar(path, meta):
    with tarfile.open(path, "w") as tar:
        def put(name, data):
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
        put("audio_metadata.json", json.dumps(meta, ensure_ascii=False).encode("utf-8"))
        for f in TAR_FILES:
            put("cleaned_audios/" + f, ("CLIP:" + f).encode())


def run_select(select_fn, make_obj, tmp):
    """select_fn(obj, **args) over the grid; make_obj(meta_key, cfg_dict) -> the object to call it on."""
    out = []
    open(os.path.join(tmp, "ref.wav"), "wb").write(b"RIFF")
    for c in select_cases():
        args = {k: (v.replace("{TMP}", tmp) if isinstance(v, str) else v) for k, v in c["args"].items()}
        try:
            audio, text = select_fn(make_obj(c["meta"], c["cfg"]), **args)
            if isinstance(audio, (bytes, bytearray)):
                res = {"ok": ["bytes", bytes(audio).decode("latin1"), text]}
            else:
                res = {"ok": ["path", str(audio).replace(tmp, "{TMP}"), text]}
        except Exception as e:  # noqa: BLE001  (the exception type is the datum)
            res = {"err": [type(e).__name__, str(e).replace(tmp, "{TMP}")]}
        out.append({**c, "result": res})
    return out


class CountingSession:
    """Stands where onnxruntime.InferenceSession stands for the transformer graph: returns (noise + 1, time_step + fuse)."""

    def __init__(self, fuse):
        self.fuse, self.calls = fuse, []

    def run(self, output_names, feed):
        keys = list(feed.keys())
        ts = feed[keys[7]]
        self.calls.append({"outputs": list(output_names), "feed_keys": keys, "time_step": int(np.asarray(ts).reshape(-1)[0]),
                           "noise0": float(np.asarray(feed[keys[0]]).reshape(-1)[0])})
        return [np.asarray(feed[keys[0]]) + 1.0, (np.asarray(ts) + self.fuse).astype(np.int32)]


STEP_GRID = [(n, f) for n, f in itertools.product([2, 3, 8, 32, 33], [1, 2, 3, 5])]
IN_NAMES = ["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k", "cat_mel_text", "cat_mel_text_drop", "time_step"]
OUT_NAMES = ["denoised", "time_step_out"]


def run_step_pattern(run_steps_fn):
    """run_steps_fn(fake_self, *8 arrays) -> (noise, time_step), for every (nfe_step, fuse_nfe) of the grid."""
    out = []
    for nfe, fuse in STEP_GRID:
        sess = CountingSession(fuse)
        msm = types.SimpleNamespace(sessions={"transformer": sess}, input_names={"transformer": IN_NAMES}, output_names={"transformer": OUT_NAMES})
        fake = types.SimpleNamespace(config=types.SimpleNamespace(nfe_step=nfe, fuse_nfe=fuse), model_session_manager=msm)
        arrs = [np.full((1, 2, 3), 10.0 * i, dtype=np.float32) for i in range(7)] + [np.array([0], dtype=np.int32)]
        noise, ts = run_steps_fn(fake, *arrs)
        out.append({"nfe_step": nfe, "fuse_nfe": fuse, "n_calls": len(sess.calls), "time_steps": [c["time_step"] for c in sess.calls],
                    "noise_in": [c["noise0"] for c in sess.calls], "feed_keys": sess.calls[0]["feed_keys"] if sess.calls else [],
                    "outputs": sess.calls[0]["outputs"] if sess.calls else [], "final_noise0": float(np.asarray(noise).reshape(-1)[0]),
                    "final_time_step": int(np.asarray(ts).reshape(-1)[0])})
    return out


# ---------------------------------------------------------------------------------------------- C / D helpers
ENGINE_CFG = dict(synthetic_model=True, model_spec="tiny", nfe_step=4, max_chunk_duration=8.0)
NOISE_SEED = 2024
SYNTH_CASES = [
    {"name": "short_default", "kwargs": {"text": "Xin chào các bạn."}},
    {"name": "filters", "kwargs": {"text": "Tạm biệt nhé!", "gender": "male", "group": "news", "area": "southern", "emotion": "serious"}},
    {"name": "iteration", "kwargs": {"text": "Một hai ba.", "sample_iteration": 1}},
    {"name": "long_chunks", "kwargs": {"text": "Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ nhé. " * 3}},
    {"name": "to_file", "kwargs": {"text": "Lưu vào tệp.", "output_path": "{TMP}/out/c.wav"}},
]


def oracle_factory(spec, weights, config):
    from oracle.vv_oracle import Oracle, OracleSession
    orc = Oracle(spec, weights, nfe_step=config.nfe_step)
    return {k: OracleSession(orc, k, seed=config.random_seed) for k in ("preprocess", "transformer", "decode")}


def reseed(msm, seed=NOISE_SEED):
    import torch
    for s in msm.sessions.values():
        s.gen = torch.Generator().manual_seed(seed)


def pcm_record(arrays, key, wave):
    wave = np.asarray(wave).reshape(-1)
    assert wave.dtype == np.int16
    arrays[key] = wave
    return {"key": key, "n": int(wave.size), "sha1": hashlib.sha1(wave.tobytes()).hexdigest()}


def main():
    ref = mhg.load_reference()
    gold, arrays = {}, {}
    tmp = tempfile.mkdtemp(prefix="vvgold_")

    # ---- A. select_sample
    ref_msm_cls = ref["model"].ModelSessionManager
    tars = {}
    for k, meta in SELECT_META.items():
        tars[k] = os.path.join(tmp, f"sel_{k}.tar")
        write_select_tar(tars[k], meta)

    def make_ref_obj(meta_key, cfg):
        o = object.__new__(ref_msm_cls)                   # no __init__: it would ask onnxruntime for providers
        o.config = types.SimpleNamespace(**cfg, ensure_model_downloaded=lambda p=tars[meta_key]: p)
        o.sample_metadata = SELECT_META[meta_key]
        o.temp_dir = None
        return o
    gold["select_meta"] = SELECT_META
    gold["select_tar_files"] = TAR_FILES
    gold["select_sample"] = run_select(ref_msm_cls.select_sample, make_ref_obj, tmp)

    # ---- B. transformer step loop
    ref_eng_cls = ref["tts_engine"].TTSEngine
    gold["step_pattern"] = run_step_pattern(ref_eng_cls._run_transformer_steps)

    # ---- C. the reference's synthesize() over this repo's session manager with CPU sessions
    from vietvoice_tts_amd.core import AudioProcessor as OurAP, ModelConfig as OurCfg, TTSEngine as OurEngine
    cache = os.path.join(tmp, "models")
    cfg = OurCfg(model_cache_dir=cache, **ENGINE_CFG)
    ours = OurEngine(cfg, session_factory=oracle_factory)
    msm = ours.model_session_manager

    class RefAudio(ref["audio_processor"].AudioProcessor):
        @staticmethod
        def load_audio(path_or_bytes, sample_rate):       # pydub is absent: this repo's RIFF loader supplies the samples
            return OurAP.load_audio(path_or_bytes, sample_rate)
    # soundfile is absent as well: the reference's save_audio keeps its own checks and calls sf.write, which lands in this repo's writer
    sys.modules["soundfile"].write = lambda path, data, sr, format=None: OurAP.save_audio(np.asarray(data), str(path), sr)
    fake = types.SimpleNamespace(config=cfg, model_session_manager=msm, text_processor=ref["text_processor"].TextProcessor(msm.vocab_path),
                                 audio_processor=RefAudio(), sample_cache={})
    for name in ("_prepare_inputs", "_run_preprocess", "_run_transformer_steps", "_run_decode"):
        setattr(fake, name, types.MethodType(getattr(ref_eng_cls, name), fake))
    synth = []
    for c in SYNTH_CASES:
        kw = {k: (v.replace("{TMP}", tmp) if isinstance(v, str) else v) for k, v in c["kwargs"].items()}
        reseed(msm)
        wave_ref, secs = ref_eng_cls.synthesize(fake, **kw)
        reseed(msm)
        wave_ours, _ = ours.synthesize(**kw)
        assert np.array_equal(np.asarray(wave_ref).reshape(-1), wave_ours), c["name"]      # same machine, same sessions: identical
        rec = {"name": c["name"], "kwargs": c["kwargs"], "pcm": pcm_record(arrays, "synth_" + c["name"], wave_ref), "seconds_positive": secs > 0}
        if "output_path" in kw:
            data = open(kw["output_path"], "rb").read()
            rec["file"] = {"riff": data[:4].decode("latin1"), "wave": data[8:12].decode("latin1"), "size": len(data)}
        synth.append(rec)
    gold["synthesize"] = synth
    gold["engine_cfg"] = ENGINE_CFG
    gold["noise_seed"] = NOISE_SEED

    # ---- D. the reference client over vietvoice_tts_amd.core
    import vietvoice_tts_amd.core as our_core
    import vietvoice_tts_amd.core.model_config as our_mc
    pkg = types.ModuleType("refclientpkg")
    pkg.__path__ = [REF_PKG]
    sys.modules["refclientpkg"] = pkg
    sys.modules["refclientpkg.core"] = our_core
    sys.modules["refclientpkg.core.model_config"] = our_mc
    spec = importlib.util.spec_from_file_location("refclientpkg.client", os.path.join(REF_PKG, "client.py"))
    client = importlib.util.module_from_spec(spec)
    sys.modules["refclientpkg.client"] = client
    spec.loader.exec_module(client)
    assert client.TTSEngine is OurEngine and client.ModelConfig is OurCfg and client.MODEL_GENDER == ["male", "female"]

    received = []

    class RecordingEngine(OurEngine):
        def __init__(self, config=None):
            super().__init__(config, session_factory=oracle_factory)
            reseed(self.model_session_manager)

        def synthesize(self, *a, **k):
            assert not a, "the reference client passes keyword arguments only"
            received.append({kk: (vv.replace(tmp, "{TMP}") if isinstance(vv, str) else vv) for kk, vv in k.items()})
            reseed(self.model_session_manager)
            return super().synthesize(**k)
    client.TTSEngine = RecordingEngine                    # what `TTSApi.engine` instantiates (client.py:34-39)
    calls = []
    with client.TTSApi(cfg) as api:
        def do(method, **kw):
            n0 = len(received)
            real = {k: (v.replace("{TMP}", tmp) if isinstance(v, str) else v) for k, v in kw.items()}
            try:
                res = getattr(api, method)(**real)
                if method == "synthesize":
                    out = {"pcm": pcm_record(arrays, f"client_{len(calls)}", res[0]), "seconds_positive": res[1] > 0, "tuple_len": len(res)}
                elif method == "synthesize_to_file":
                    data = open(real["output_path"], "rb").read()
                    out = {"returns_float": isinstance(res, float), "positive": res > 0, "file": {"riff": data[:4].decode("latin1"), "size": len(data)}}
                else:
                    out = {"value": res}
            except Exception as e:  # noqa: BLE001
                out = {"err": [type(e).__name__, str(e)]}
            calls.append({"method": method, "api_kwargs": kw, "engine_kwargs": received[n0:], "result": out})
        do("synthesize", text="Xin chào từ client.")
        do("synthesize", text="Giọng nam đọc tin.", gender="male", group="news", area="southern", emotion="serious", sample_iteration=0)
        do("synthesize_to_file", text="Ghi ra tệp.", output_path="{TMP}/out/client.wav", sample_iteration=1)
        do("synthesize", text=None)
        do("synthesize", text="x", gender="robot")
        do("validate_configuration")
        engine_before = api._engine
    calls.append({"method": "__exit__", "engine_cleaned": engine_before.model_session_manager.vocab_path is None})
    n0 = len(received)
    secs = client.synthesize("Hàm tiện ích.", os.path.join(tmp, "out", "conv.wav"), config=cfg, gender="female")
    data = open(os.path.join(tmp, "out", "conv.wav"), "rb").read()
    calls.append({"method": "module.synthesize", "api_kwargs": {"text": "Hàm tiện ích.", "output_path": "{TMP}/out/conv.wav", "gender": "female"},
                  "engine_kwargs": received[n0:], "result": {"returns_float": isinstance(secs, float), "file": {"riff": data[:4].decode("latin1"), "size": len(data)}}})
    gold["client"] = calls
    ours.cleanup()

    with open(os.path.join(HERE, "api_golden.json"), "w", encoding="utf-8") as f:
        json.dump(gold, f, ensure_ascii=False, indent=1)
    np.savez_compressed(os.path.join(HERE, "api_golden.npz"), **arrays)
    print("select cases", len(gold["select_sample"]), "| step grid", len(gold["step_pattern"]), "| synth", len(synth), "| client calls", len(calls),
          "| arrays", {k: v.size for k, v in arrays.items()})


if __name__ == "__main__":
    main()

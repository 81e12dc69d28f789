"""Generates tests/golden/host_golden.json + host_golden.npz from the REFERENCE's own host helpers.

Run in the build container only (the reference is mounted read-only at /root/reference and never
travels to the GPU box):   python tests/golden/make_host_golden.py

Technique: vietvoicetts/core/{text_processor,audio_processor,tts_engine}.py
are loaded BY FILE PATH as a throw-away package with stub modules standing in for the absent
third-party imports (loguru, soundfile, pydub, onnxruntime).  Only pure-numpy/regex methods are
executed.  Nothing that touches the network is called: no ModelConfig(), no TTSApi(), no
ensure_model_downloaded(); the engine's _prepare_inputs is driven with a SimpleNamespace config
and an in-memory int16 clip in place of the pydub decoder.  The fixtures are data (inputs and
expected outputs); no reference source text is stored.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/vietvoicetts/core"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    class _Log:
        def __getattr__(self, _n):
            return lambda *a, **k: None
    sys.modules.setdefault("loguru", types.SimpleNamespace(logger=_Log()))
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    pd = types.ModuleType("pydub")
    pd.AudioSegment = object
    sys.modules.setdefault("pydub", pd)
    class _Any(types.ModuleType):
        def __getattr__(self, _n):
            if _n.startswith("__"):                    # module introspection (__file__, __path__, ...) must not see a stub
                raise AttributeError(_n)
            return type("Stub", (), {})
    sys.modules.setdefault("onnxruntime", _Any("onnxruntime"))
    pkg = types.ModuleType("refcore")
    pkg.__path__ = [REF]
    sys.modules["refcore"] = pkg
    mods = {}
    for name in ("model_config", "text_processor", "audio_processor", "model", "tts_engine"):
        spec = importlib.util.spec_from_file_location(f"refcore.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refcore.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


VOCAB = [" "] + list("abcdefghijklmnopqrstuvwxyz") + list("àáảãạăằắẳẵặâầấẩẫậèéẻẽẹêềếểễệđìíỉĩịòóỏõọôồốổỗộơờớởỡợùúủũụưừứửữựỳỵỷỹý") + list(".,!?'") + list("ABCDXYZ0123456789")

CLEAN_CASES = [
    "  a;b:c(d)   efg! ", "Xin chào, thế giới.", "Hello\nworld\n\nthird line.", "", "   ", "no punctuation at end",
    "weird #chars* and ~tildes~ [brackets] {braces} <tags>", "Nhiều.... dấu,,,, câu;;; liên:: tiếp (ngoặc)",
    "ĐÂY LÀ CHỮ HOA? Đúng vậy!", "tabs\tand\r\nwindows newlines", "số 123 và 45,6%. email@test.com $5 & more/less",
    "kết thúc bằng dấu phẩy,", "emoji 😀 và ký tự lạ ©®", "a\nb.\nc",
]
LEN_CASES = [("a, b, c.", r"[,.]"), ("Xin chào, thế giới.", r".,?!:"), ("", r".,?!:"), ("hello world", r".,?!:"),
             ("a.b,c?d!e:f", r"[.,?!:]"), ("tiếng Việt có dấu, rất nhiều dấu.", r".,?!:"), ("x.,?!:y", r".,?!:")]
CHUNK_CASES = [
    ("This is a long sentence. This is another long sentence. And a third one.", 30),
    ("This is a long sentence. This is another long sentence. And a third one.", 135),
    ("", 50), ("   ", 50), ("Supercalifragilisticexpialidocious", 10), ("one two three four five six seven eight nine ten", 12),
    ("Câu một rất ngắn. Câu hai cũng ngắn! Câu ba thì dài hơn một chút, có dấu phẩy, và thêm vài từ nữa? Hết.", 40),
    ("a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p", 9), ("Hi. Yo. This is a considerably longer closing sentence here.", 45),
    ("word " * 60, 50), ("First part is long enough here, second, third part also fairly long. Ok.", 25),
    ("Một hai ba bốn năm sáu bảy tám chín mười. " * 6, 80), ("A.B.C. D!E? F", 5), ("Tail short. Hi", 40),
]


def main():
    ref = load_reference()
    tmp_vocab = os.path.join(OUT, "_vocab_tmp.txt")
    with open(tmp_vocab, "w", encoding="utf-8") as f:
        f.write("\n".join(VOCAB) + "\n")
    tp = ref["text_processor"].TextProcessor(tmp_vocab)
    ap = ref["audio_processor"].AudioProcessor
    gold = {"vocab": VOCAB, "vocab_size": tp.vocab_size}
    gold["clean_text"] = [[s, tp.clean_text(s)] for s in CLEAN_CASES]
    gold["text_length"] = [[s, p, tp.calculate_text_length(s, p)] for s, p in LEN_CASES]
    gold["chunk_text"] = [[s, m, tp.chunk_text(s, m)] for s, m in CHUNK_CASES]
    idx_cases = [list("xin chào"), list("Zebra? 9!"), list("☃ unknown ☃")]
    gold["text_to_indices"] = [["".join(c), tp.text_to_indices([c]).tolist()] for c in idx_cases]

    arrays = {}
    rng = np.random.default_rng(9527)
    norm_in = [np.array([0, .5, -.5, 1, -1], dtype=np.float32), (rng.standard_normal(4000) * 5000 + 300).astype(np.float32),
               np.zeros(16, dtype=np.float32), (rng.standard_normal(777) * 0.01).astype(np.float32),
               np.full(100, 12345.0, dtype=np.float32)]
    for i, a in enumerate(norm_in):
        arrays[f"norm_in_{i}"] = a
        arrays[f"norm_out_{i}"] = ap.normalize_to_int16(a)
    gold["n_norm"] = len(norm_in)
    clip_in = [np.array([0, 32767, -32768, 100], dtype=np.int16), np.array([1.0, np.nan, np.inf, -np.inf, 40000.0], dtype=np.float32),
               (rng.standard_normal(500) * 3000).astype(np.int16), np.array([5.0, np.nan, -7.0], dtype=np.float32)]
    for i, a in enumerate(clip_in):
        arrays[f"clip_in_{i}"] = a
        arrays[f"clip_out_{i}"] = np.asarray(ap.fix_clipped_audio(a))
    gold["n_clip"] = len(clip_in)
    xf_cases = []
    waves_sets = [
        [np.full(16000, 1000, dtype=np.int16), np.full(16000, 2000, dtype=np.int16)],
        [(rng.standard_normal(n) * a).astype(np.int16) for n, a in ((9000, 3000), (7000, 800), (12000, 5000))],
        [(rng.standard_normal(n) * a).astype(np.int16).reshape(1, 1, -1) for n, a in ((3000, 2000), (500, 50), (4000, 2500))],
        [(rng.standard_normal(2500) * 4000).astype(np.int16)],
        [np.array([32767, -32768] * 2000, dtype=np.int16), (rng.standard_normal(5000) * 1000).astype(np.int16)],
    ]
    for i, ws in enumerate(waves_sets):
        for sr, dur in ((16000, 0.1), (24000, 0.1), (24000, 0.0), (24000, 1.0)):
            key = f"xf_{i}_{sr}_{int(dur * 1000)}"
            for j, w in enumerate(ws):
                arrays[f"{key}_in_{j}"] = w
            arrays[f"{key}_plain"] = np.asarray(ap.concatenate_with_crossfade([w.copy() for w in ws], dur, sr))
            arrays[f"{key}_improved"] = np.asarray(ap.concatenate_with_crossfade_improved([w.copy() for w in ws], dur, sr))
            xf_cases.append([key, len(ws), sr, dur])
    gold["crossfade"] = xf_cases
    gold["crossfade_empty_len"] = int(np.asarray(ap.concatenate_with_crossfade_improved([], 0.1, 24000)).size)

    # ---- _prepare_inputs (duration model, chunk plan, frames rule) with an in-memory reference clip
    eng_cls = ref["tts_engine"].TTSEngine
    prep = []
    long_text = ("Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ nhé. " * 14).strip()
    cases = [
        dict(S=144000, ref_text="xin chào các bạn, đây là giọng đọc mẫu của tôi.", text="Tôi rất vui được gặp bạn hôm nay.", speed=0.9, max_chunk=20.0),
        dict(S=144000, ref_text="xin chào các bạn, đây là giọng đọc mẫu của tôi.", text=long_text, speed=0.9, max_chunk=20.0),
        dict(S=72000, ref_text="ngắn thôi.", text=long_text, speed=1.3, max_chunk=15.0),
        dict(S=48000, ref_text="a b c", text="x", speed=0.9, max_chunk=20.0),
        dict(S=120001, ref_text="mẫu tham chiếu dài vừa phải, có dấu phẩy.", text="Một câu. Hai câu! Ba câu? " * 9, speed=0.5, max_chunk=12.0),
    ]
    for c in cases:
        clip = (rng.standard_normal(c["S"]) * 2000).astype(np.int16)

        class FakeAudio:
            @staticmethod
            def load_audio(_p, _sr, clip=clip):
                return clip
        cfg = types.SimpleNamespace(sample_rate=24000, hop_length=256, pause_punctuation=r".,?!:", speed=c["speed"],
                                    min_target_duration=1.0, max_chunk_duration=c["max_chunk"])
        fake = types.SimpleNamespace(config=cfg, text_processor=tp, audio_processor=FakeAudio)
        res = eng_cls._prepare_inputs(fake, "in-memory", c["ref_text"], c["text"])
        prep.append(dict(case=c, n_chunks=len(res), max_duration=[int(r[2][0]) for r in res],
                         text_ids=[r[1].tolist() for r in res], audio_shape=list(res[0][0].shape),
                         time_step=[int(r[3][0]) for r in res], dtypes=[str(res[0][i].dtype) for i in range(4)]))
    gold["prepare_inputs"] = prep
    os.remove(tmp_vocab)
    with open(os.path.join(OUT, "host_golden.json"), "w", encoding="utf-8") as f:
        json.dump(gold, f, ensure_ascii=False, indent=1)
    np.savez_compressed(os.path.join(OUT, "host_golden.npz"), **arrays)
    print("wrote", len(gold), "groups,", len(arrays), "arrays")


if __name__ == "__main__":
    main()

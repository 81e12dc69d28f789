"""Model pack: the single-file container ModelSessionManager opens, mirroring the layout of the
reference's ``model-bin.pt`` tar (vietvoicetts/core/model.py:73-77,84,109,207):

    audio_metadata.json            list of {file_name, gender, group, area, emotion, text}
    vocab.txt                      one symbol per line; line number = token id
    cleaned_audios/<file_name>     built-in reference clips (WAV here)
    model_spec.json                architecture constants + where the weights come from   (ours)
    weights.safetensors            optional fp32 tensors in torch-native layouts            (ours)

The reference archive carries three ONNX graphs instead of the last two entries; importing their
initializers is SURVEY.md 8(f) N1 ("next").  With no network and no checkpoint, the weights of a
synthetic pack are regenerated from (spec, seed) at load time instead of being stored (1.3 GB).
"""
from __future__ import annotations

import io
import json
import math
import tarfile
from typing import Dict, List, Tuple

import numpy as np
import torch

from .model_spec import ModelSpec, make_synthetic_weights

_VOICES = [
    ("female", "audiobook", "northern", "neutral", "xin chào các bạn, đây là giọng đọc mẫu dùng để tham chiếu."),
    ("female", "story", "northern", "happy", "ngày xửa ngày xưa, ở một ngôi làng nhỏ bên bờ sông."),
    ("male", "news", "southern", "serious", "bản tin thời sự hôm nay có những nội dung chính sau đây."),
    ("male", "audiobook", "central", "neutral", "chương một, buổi sáng hôm ấy trời trong và gió nhẹ."),
    ("female", "interview", "southern", "surprised", "thật vậy sao, tôi chưa từng nghe điều đó bao giờ."),
    ("female", "audiobook", "northern", "neutral", "mẫu thứ hai cho cùng một bộ lọc giọng đọc mặc định."),
]

_SYMBOLS = (" " + "abcdefghijklmnopqrstuvwxyz" + "àáảãạăằắẳẵặâầấẩẫậèéẻẽẹêềếểễệđìíỉĩịòóỏõọôồốổỗộơờớởỡợùúủũụưừứửữựỳỵỷỹý"
            + ".,!?'" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "0123456789" + "ÀÁẢÃẠĂÂÈÉÊĐÌÍÒÓÔƠÙÚƯÝ" + "@$%&/:;()-\"")


def spec_by_name(name: str) -> ModelSpec:
    return {"full": ModelSpec.full, "small": ModelSpec.small, "tiny": ModelSpec.tiny}[name]()


def synthetic_voice(seed: int, seconds: float, sample_rate: int = 24000) -> np.ndarray:
    """Band-limited noise clip (sum of sinusoids), peak-normalised like the reference loader."""
    g = torch.Generator().manual_seed(seed)
    n = int(seconds * sample_rate)
    t = torch.arange(n, dtype=torch.float64) / sample_rate
    f = 80.0 + 7520.0 * torch.rand(32, generator=g, dtype=torch.float64)
    ph = 2 * math.pi * torch.rand(32, generator=g, dtype=torch.float64)
    x = torch.sin(2 * math.pi * f[:, None] * t[None, :] + ph[:, None]).sum(0)
    x = x - x.mean()
    return (x * (29491.0 / x.abs().max())).to(torch.int16).numpy()


def _wav_bytes(pcm: np.ndarray, rate: int) -> bytes:
    from .core.audio_processor import AudioProcessor
    return AudioProcessor.to_wav_bytes(pcm, rate)


def write_synthetic_pack(path: str, spec_name: str = "full", seed: int = 9527) -> None:
    spec = spec_by_name(spec_name)
    vocab = list(dict.fromkeys(_SYMBOLS))[: spec.vocab_size]
    meta, clips = [], []
    for i, (gender, group, area, emotion, text) in enumerate(_VOICES):
        name = f"sample_{i:03d}.wav"
        meta.append({"file_name": name, "gender": gender, "group": group, "area": area, "emotion": emotion, "text": text})
        clips.append((name, _wav_bytes(synthetic_voice(seed + i, 3.0 + 0.5 * i), spec.sample_rate)))
    entries: List[Tuple[str, bytes]] = [
        ("audio_metadata.json", json.dumps(meta, ensure_ascii=False).encode("utf-8")),
        ("vocab.txt", ("\n".join(vocab) + "\n").encode("utf-8")),
        ("model_spec.json", json.dumps({"spec": json.loads(spec.to_json()), "weights": {"kind": "synthetic", "seed": seed}}).encode()),
    ] + [("cleaned_audios/" + n, b) for n, b in clips]
    with tarfile.open(path, "w") as tar:
        for name, data in entries:
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))


def read_pack_model(tar: tarfile.TarFile) -> Tuple[ModelSpec, Dict[str, torch.Tensor]]:
    """-> (spec, fp32 weights in torch-native layouts)."""
    names = tar.getnames()
    if "model_spec.json" not in names:
        if any(n.endswith(".onnx") for n in names):        # the reference's own layout (core/model.py:73-110): read the initializers
            from .onnx_import import import_archive
            return import_archive(tar)
        raise FileNotFoundError("Model file 'model_spec.json' not found in model archive")
    doc = json.load(tar.extractfile("model_spec.json"))
    spec = ModelSpec.from_json(json.dumps(doc["spec"]))
    src = doc.get("weights", {"kind": "synthetic", "seed": 9527})
    if src["kind"] == "synthetic":
        return spec, make_synthetic_weights(spec, int(src["seed"]))
    if src["kind"] == "safetensors":
        from safetensors.torch import load as st_load
        return spec, st_load(tar.extractfile(src.get("file", "weights.safetensors")).read())
    raise RuntimeError(f"unknown weight source {src['kind']!r}")

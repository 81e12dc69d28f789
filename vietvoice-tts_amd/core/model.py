"""ModelSessionManager -- the drop-in boundary (reference vietvoicetts/core/model.py:18-224).

Same surface: ``sessions`` / ``input_names`` / ``output_names`` dicts keyed 'preprocess' /
'transformer' / 'decode' (:24-26,104-106), ``vocab_path`` (:123), ``sample_metadata`` (:84),
``load_models()`` (:131-135), ``select_sample(...)`` with the reference's selection and error
semantics (:137-214), ``cleanup()`` (:216-221).  Behind it, instead of three onnxruntime
sessions, one HipSynth engine (hand-written gfx950 kernels behind the C ABI); the three session
objects keep onnxruntime's ``run(output_names, feed)`` shape with the reference's positional I/O
order (core/tts_engine.py:140-144,161-170,182-185,229-230) so reference-style callers work, while
TTSEngine uses the device-resident batched path directly.

There is no CPU execution provider: without the HIP library and a GPU, load_models() raises.
(A ``session_factory`` can be injected -- the CPU plumbing tests pass the oracle's sessions.)
"""
from __future__ import annotations

import json
import logging
import random
import shutil
import tarfile
import tempfile
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from .model_config import MODEL_AREA, MODEL_EMOTION, MODEL_GENDER, MODEL_GROUP, ModelConfig

logger = logging.getLogger("vietvoicetts")

SESSION_IO = {
    "preprocess": (["audio", "text_ids", "max_duration"],
                   ["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k", "cat_mel_text", "cat_mel_text_drop", "ref_signal_len"]),
    "transformer": (["noise", "rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k", "cat_mel_text", "cat_mel_text_drop", "time_step"],
                    ["denoised", "time_step_out"]),
    "decode": (["denoised", "ref_signal_len"], ["output_audio"]),
}


class _IoName:
    def __init__(self, name: str):
        self.name = name


class HipSession:
    """One stage of the HIP engine with onnxruntime.InferenceSession's call shape (batch 1, host
    numpy in and out -- the reference's contract, including its host round trips)."""

    def __init__(self, engine, kind: str, noise_gen, fuse_nfe: int = 1):
        self.engine, self.kind, self.noise_gen, self.fuse_nfe = engine, kind, noise_gen, max(1, int(fuse_nfe))
        self._in, self._out = SESSION_IO[kind]

    def get_inputs(self):
        return [_IoName(n) for n in self._in]

    def get_outputs(self):
        return [_IoName(n) for n in self._out]

    def run(self, output_names, feed: Dict[str, np.ndarray]) -> List[np.ndarray]:
        import torch
        eng = self.engine
        dev = eng.device
        vals = [feed[n] for n in self._in]
        if self.kind == "preprocess":
            audio = torch.from_numpy(np.ascontiguousarray(np.asarray(vals[0]).reshape(1, -1))).to(torch.int16).to(dev)
            n_audio = audio.shape[1]
            if n_audio < eng.spec.n_fft:                       # plane at least n_fft wide (vv_preprocess's contract); the tail is never read
                audio = torch.nn.functional.pad(audio, (0, eng.spec.n_fft - n_audio))
            ids = torch.from_numpy(np.ascontiguousarray(np.asarray(vals[1]).reshape(1, -1))).to(torch.int32).to(dev)
            n = int(np.asarray(vals[2]).reshape(-1)[0])
            i32 = lambda v: torch.tensor([v], dtype=torch.int32, device=dev)
            pre = eng.preprocess(audio, i32(n_audio), ids, i32(ids.shape[1]), i32(n), n, audio_len_host=[n_audio])
            noise = torch.randn((1, n, eng.spec.n_mel), generator=self.noise_gen, dtype=torch.float32)
            res = {"noise": noise.numpy(), "ref_signal_len": pre["ref_signal_len"].cpu().numpy().astype(np.int64)}
            for k in ("rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k"):
                res[k] = pre[k].cpu().numpy()[None]
            for k in ("cat_mel_text", "cat_mel_text_drop"):
                res[k] = pre[k].cpu().numpy()
            return [res[n_] for n_ in self._out]
        if self.kind == "transformer":
            x = torch.from_numpy(np.ascontiguousarray(vals[0], dtype=np.float32)).to(dev)
            n = x.shape[1]
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            pre = {"rope_cos_q": up(vals[1]).reshape(n, -1), "rope_sin_q": up(vals[2]).reshape(n, -1),
                   "rope_cos_k": up(vals[3]).reshape(n, -1), "rope_sin_k": up(vals[4]).reshape(n, -1),
                   "cat_mel_text": up(vals[5]), "cat_mel_text_drop": up(vals[6]),
                   "seq_len": torch.tensor([n], dtype=torch.int32, device=dev)}
            # The bf16 model computes the standard rope angles instead of reading the tables (vv_set_rope_theta).  This session takes
            # its tables from the CALLER (the reference's I/O contract, core/tts_engine.py:161-170): whenever they are not the standard
            # ones this engine produces in its preprocess stage, THIS call reads them -- never ignore what was fed, and never change
            # the mode for other sessions of the engine.  Checked on every call (four small device compares).
            same = n <= eng.rope[0].shape[0] and all(pre[k_].shape == eng.rope[i][:n].shape and torch.equal(pre[k_], eng.rope[i][:n])
                                                     for i, k_ in enumerate(("rope_cos_q", "rope_sin_q", "rope_cos_k", "rope_sin_k")))
            step = int(np.asarray(vals[7]).reshape(-1)[0])
            k = min(self.fuse_nfe, eng.n_steps - step)
            if same:
                eng.transformer_steps(x, pre, step, k)
            else:
                if not getattr(self, "_rope_warned", False):
                    logger.warning("transformer session: non-standard rope tables fed; the tables are read instead of computed")
                    self._rope_warned = True
                with eng.reading_rope_tables():
                    eng.transformer_steps(x, pre, step, k)
            return [x.cpu().numpy(), np.array([step + k], dtype=np.int32)]
        x = torch.from_numpy(np.ascontiguousarray(vals[0], dtype=np.float32)).to(dev)
        n = x.shape[1]
        ref_len = int(np.asarray(vals[1]).reshape(-1)[0])
        pre = {"ref_signal_len": torch.tensor([ref_len], dtype=torch.int32, device=dev),
               "seq_len": torch.tensor([n], dtype=torch.int32, device=dev)}
        pcm, pcm_len = eng.decode(x, pre, max(n - ref_len, 1))
        return [pcm[:, : int(pcm_len[0])].cpu().numpy().reshape(1, 1, -1)]


class ModelSessionManager:
    def __init__(self, config: ModelConfig, session_factory: Optional[Callable] = None):
        self.config = config
        self._session_factory = session_factory
        self.providers = self._get_optimal_providers()
        self.sessions: Dict[str, object] = {}
        self.input_names: Dict[str, List[str]] = {}
        self.output_names: Dict[str, List[str]] = {}
        self.sample_metadata = {}
        self.temp_dir = None
        self.vocab_path = None
        self.engine = None           # HipSynth when the HIP sessions are active
        self.noise_gen = None
        self.spec = None
        self._clip_cache: Dict[str, bytes] = {}

    def _get_optimal_providers(self) -> List[str]:
        """The reference ranks onnxruntime providers (model.py:31-48); this build has exactly one."""
        return ["HIPExecutionProvider"] if self._session_factory is None else ["InjectedSessionProvider"]

    def _load_models_from_file(self) -> None:
        model_path = self.config.ensure_model_downloaded()
        if not Path(model_path).exists():
            raise FileNotFoundError(f"Model file not found: {model_path}")
        try:
            import torch
            from ..model_pack import read_pack_model
            with tarfile.open(model_path, "r") as tar:
                members = tar.getnames()
                self.sample_metadata = json.load(tar.extractfile("audio_metadata.json"))
                spec, weights = read_pack_model(tar)
                vocab_member = next((m for m in members if m.endswith("vocab.txt")), None)
                if not vocab_member:
                    raise FileNotFoundError("Vocabulary file 'vocab.txt' not found in model archive")
                self.temp_dir = tempfile.mkdtemp(prefix="tts_vocab_")
                vocab_tmp = Path(self.temp_dir) / "vocab.txt"
                with open(vocab_tmp, "wb") as fh:
                    fh.write(tar.extractfile(vocab_member).read())
                self.vocab_path = str(vocab_tmp)
            self.spec = spec
            self.noise_gen = torch.Generator().manual_seed(self.config.random_seed)
            if self._session_factory is not None:
                made = self._session_factory(spec, weights, self.config)
            else:
                from ..runtime import HipSynth
                self.engine = HipSynth(spec, weights, device=self.config.device, acoustic_dtype=self.config.acoustic_dtype,
                                       nfe_step=self.config.nfe_step)
                made = {k: HipSession(self.engine, k, self.noise_gen, self.config.fuse_nfe) for k in SESSION_IO}
            for name in ("preprocess", "transformer", "decode"):
                sess = made[name]
                self.sessions[name] = sess
                self.input_names[name] = [i.name for i in sess.get_inputs()]
                self.output_names[name] = [o.name for o in sess.get_outputs()]
        except Exception as e:
            if self.temp_dir and Path(self.temp_dir).exists():
                shutil.rmtree(self.temp_dir)
                self.temp_dir = None
            raise RuntimeError(f"Failed to load models from file: {str(e)}")

    def load_models(self) -> None:
        random.seed(self.config.random_seed)
        self._load_models_from_file()

    def select_sample(self, gender: Optional[str] = None, group: Optional[str] = None, area: Optional[str] = None,
                      emotion: Optional[str] = None, sample_iteration: Optional[int] = None,
                      reference_audio: Optional[str] = None, reference_text: Optional[str] = None) -> Tuple[object, str]:
        """-> (reference audio path or WAV bytes, reference text); semantics of model.py:137-214."""
        # falsy arguments fall back to the config defaults, and the defaulted values are what the error text reports (:147-150,213)
        gender, group = gender or self.config.gender, group or self.config.group
        area, emotion = area or self.config.area, emotion or self.config.emotion
        wanted = {}
        for key, value, allowed in (("gender", gender, MODEL_GENDER), ("group", group, MODEL_GROUP),
                                    ("area", area, MODEL_AREA), ("emotion", emotion, MODEL_EMOTION)):
            if value is not None:
                if value not in allowed:
                    raise ValueError(f"Invalid {key}: {value}. Must be one of {allowed}")
                wanted[key] = value
        if reference_audio is not None:
            if reference_text is None:
                raise ValueError("Reference text is required when using reference audio")
            if not Path(reference_audio).exists():
                raise FileNotFoundError(f"Reference audio file not found: {reference_audio}")
            if wanted:
                raise ValueError(f"Cannot use reference audio and text with options: {list(wanted.keys())}")
            return reference_audio, reference_text
        try:
            matches = [(s, i) for i, s in enumerate(self.sample_metadata) if all(s[k] == v for k, v in wanted.items())]
            if not matches:
                sample, idx = self.sample_metadata[0], 0
            elif sample_iteration is not None:
                if sample_iteration >= len(matches):
                    raise ValueError(f"sample_iteration {sample_iteration} is out of range. Only {len(matches)} samples available for the given filters.")
                sample, idx = matches[sample_iteration]
            else:
                sample, idx = matches[0]
            logger.info("Selected sample #%d (%s/%s/%s/%s)", idx, sample["gender"], sample["group"], sample["area"], sample["emotion"])
            name = sample["file_name"]
            if name not in self._clip_cache:           # the reference re-opens the tar per call (model.py:206)
                with tarfile.open(self.config.ensure_model_downloaded(), "r") as tar:
                    fh = tar.extractfile("cleaned_audios/" + name)
                    if not fh:
                        raise FileNotFoundError(f"Audio file {name} not found in model archive")
                    self._clip_cache[name] = fh.read()
            return self._clip_cache[name], sample["text"]
        except KeyError:
            raise ValueError(f"Sample not found for gender: {gender}, group: {group}, area: {area}, emotion: {emotion}")

    def cleanup(self) -> None:
        if self.temp_dir and Path(self.temp_dir).exists():
            shutil.rmtree(self.temp_dir)
            self.temp_dir = None
            self.vocab_path = None
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

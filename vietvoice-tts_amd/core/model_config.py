"""ModelConfig -- the knobs of the synthesis engine, field-for-field compatible with the reference
(vietvoicetts/core/model_config.py:21-63: names, defaults, validation ranges, dict round trip
:143-153, alias TTSConfig :157, MODEL_* constants :15-18).

Differences, all additive:
  * there is no network here, so nothing is ever downloaded: ``ensure_model_downloaded`` returns
    the cached model pack if it exists, builds a SYNTHETIC pack when ``synthetic_model`` (or the
    env var VIETVOICE_TTS_SYNTHETIC=1) asks for one, and otherwise raises the same RuntimeError
    family the reference raises when its download fails (model_config.py:93-100, 106-112);
  * build-only fields (device, acoustic_dtype, model_spec, ...) with defaults, so
    ``ModelConfig(**reference_dict)`` keeps working.
The onnxruntime knobs are kept as inert fields for API compatibility.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass, fields
from pathlib import Path
from typing import Optional

logger = logging.getLogger("vietvoicetts")

MODEL_GENDER = ["male", "female"]
MODEL_GROUP = ["story", "news", "audiobook", "interview", "review"]
MODEL_AREA = ["northern", "southern", "central"]
MODEL_EMOTION = ["neutral", "serious", "monotone", "sad", "surprised", "happy", "angry"]


@dataclass
class ModelConfig:
    # --- reference fields (same order, names and defaults)
    model_url: str = "https://huggingface.co/nguyenvulebinh/VietVoice-TTS/resolve/main/model-bin.pt"
    model_cache_dir: str = "models"
    model_filename: str = "model-bin.pt"
    nfe_step: int = 32
    fuse_nfe: int = 1
    sample_rate: int = 24000
    speed: float = 0.9
    random_seed: int = 9527
    hop_length: int = 256
    gender: Optional[str] = "female"
    area: Optional[str] = "northern"
    emotion: Optional[str] = "neutral"
    group: Optional[str] = "audiobook"
    pause_punctuation: str = r".,?!:"
    cross_fade_duration: float = 0.1
    max_chunk_duration: float = 20.0
    min_target_duration: float = 1.0
    log_severity_level: int = 4
    log_verbosity_level: int = 4
    inter_op_num_threads: int = 0
    intra_op_num_threads: int = 0
    enable_cpu_mem_arena: bool = True
    # --- build-only fields
    device: str = "cuda:0"
    acoustic_dtype: str = "bf16"            # "bf16" (throughput) or "fp32" (numerics configuration)
    synthetic_model: bool = False           # build a seeded synthetic model pack when none is cached
    model_spec: str = "full"                # architecture preset of a synthetic pack: full | small | tiny
    max_batch_chunks: int = 32              # chunks of one long text synthesised per GPU batch
    use_hip_graph: bool = False             # replay the vocoder step from a captured hipGraph (fixed frame buckets)
    decode_graph_cache_entries: int = 8     # captured decode graphs kept per engine (least recently used beyond that)
    decode_graph_cache_bytes: int = 16 << 30   # HBM the cache may pin (shared workspace + per-graph I/O buffers)

    def __post_init__(self):
        if not 0.1 <= self.speed <= 5.0:
            raise ValueError("Speed must be between 0.1 and 5.0")
        if not 1 <= self.nfe_step <= 100:
            raise ValueError("NFE step must be between 1 and 100")
        if self.acoustic_dtype not in ("bf16", "fp32"):
            raise ValueError("acoustic_dtype must be 'bf16' or 'fp32'")
        self.validate_paths()

    @property
    def model_path(self) -> str:
        return str(Path(self.model_cache_dir).expanduser() / self.model_filename)

    def ensure_model_downloaded(self) -> str:
        """Return the path of the cached model pack (the reference would fetch it; we cannot)."""
        path = Path(self.model_path)
        path.parent.mkdir(parents=True, exist_ok=True)
        if path.exists():
            return str(path)
        if self.synthetic_model or os.environ.get("VIETVOICE_TTS_SYNTHETIC") == "1":
            from ..model_pack import write_synthetic_pack
            logger.info("no cached model at %s: writing a seeded synthetic model pack (%s)", path, self.model_spec)
            write_synthetic_pack(str(path), self.model_spec, seed=self.random_seed)
            return str(path)
        raise RuntimeError(
            f"Failed to download model from {self.model_url}: no network access in this build; place the model pack at "
            f"{path} or set synthetic_model=True / VIETVOICE_TTS_SYNTHETIC=1")

    def validate_paths(self):
        try:
            self.ensure_model_downloaded()
        except Exception as e:
            raise RuntimeError(f"Model validation failed: {e}")

    def validate_with_reference_audio(self, reference_audio_path: str) -> bool:
        """max_chunk_duration must leave room for the clip + 1 s margin + min_target_duration
        (reference model_config.py:114-141)."""
        try:
            from .audio_processor import AudioProcessor
            ref_duration = AudioProcessor.probe_duration(reference_audio_path)
            needed = ref_duration + 1.0 + self.min_target_duration
            if self.max_chunk_duration < needed:
                logger.error("Configuration Error: reference audio %.1fs needs max_chunk_duration > %.1fs (is %.1fs)",
                             ref_duration, needed, self.max_chunk_duration)
                return False
            logger.info("Configuration valid: reference audio %.1fs, %.1fs available per chunk", ref_duration,
                        self.max_chunk_duration - ref_duration - 1.0)
            return True
        except Exception as e:      # same contract as the reference: never raises, returns False
            logger.error("Error validating reference audio: %s", e)
            return False

    @classmethod
    def from_dict(cls, config_dict: dict) -> "ModelConfig":
        return cls(**config_dict)

    def to_dict(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self)}


TTSConfig = ModelConfig   # backward-compatibility alias, as in the reference

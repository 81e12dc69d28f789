"""Host-side mirror of ``vietvoicetts.core`` (reference vietvoicetts/core/__init__.py:5-22): the same
export list, so ``vietvoicetts.client`` / the Litestar API / the CLI import it unchanged."""
from .model_config import ModelConfig, TTSConfig, MODEL_GENDER, MODEL_GROUP, MODEL_AREA, MODEL_EMOTION
from .model import ModelSessionManager
from .tts_engine import TTSEngine
from .text_processor import TextProcessor
from .audio_processor import AudioProcessor

__all__ = [
    "ModelConfig", "TTSConfig", "ModelSessionManager", "TTSEngine", "TextProcessor", "AudioProcessor",
    "MODEL_GENDER", "MODEL_GROUP", "MODEL_AREA", "MODEL_EMOTION",
]

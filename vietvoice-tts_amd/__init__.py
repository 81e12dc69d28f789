"""MI355X-native synthesis hot path for VietVoice-TTS (gfx950 HIP kernels behind the
reference's ``vietvoicetts.core`` API).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"

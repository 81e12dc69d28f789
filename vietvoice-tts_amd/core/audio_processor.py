"""AudioProcessor -- host-side audio plumbing of the hot path, behaviour-identical to the reference
(vietvoicetts/core/audio_processor.py) for every pure-numpy step: DC removal + peak 29491 + int16
truncation (:28-44), clip repair (:46-58), linear cross-fade (:69-120), RMS-matched cos^2 cross-fade
(:122-192).  Pinned by tests/golden/host_golden.npz (generated from the reference functions).

Container I/O differs by necessity: the reference decodes any format through pydub/ffmpeg (:15-26)
and writes WAVEX through soundfile (:60-67); neither exists here, so ``load_audio`` reads RIFF/WAVE
(PCM 8/16/24/32-bit, float32/64) natively, averages channels and resamples with a polyphase filter,
and ``save_audio`` writes a WAVE_FORMAT_EXTENSIBLE PCM16 file by hand.
"""
from __future__ import annotations

import io
import struct
from pathlib import Path
from typing import List, Tuple, Union

import numpy as np

_PCM_GUID = bytes.fromhex("0100000000001000800000aa00389b71")


def _parse_wav(data: bytes) -> Tuple[np.ndarray, int]:
    """-> (float32 samples scaled like int16, mono), sample_rate."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("unsupported audio container: only RIFF/WAVE can be decoded in this build (no ffmpeg)")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("malformed WAVE file: missing fmt or data chunk")
    tag, ch, rate, _br, _ba, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack("<H", fmt[24:26])[0]
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) * 256.0
        elif bits == 16:
            x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32)
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / 256.0
        elif bits == 32:
            x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float32) / 65536.0
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3:
        dt = "<f4" if bits == 32 else "<f8"
        x = np.frombuffer(pcm[: len(pcm) // (bits // 8) * (bits // 8)], dtype=dt).astype(np.float32) * 32768.0
    else:
        raise ValueError(f"unsupported WAVE format tag {tag}")
    if ch > 1:
        x = x[: len(x) // ch * ch].reshape(-1, ch).mean(axis=1)
    return np.ascontiguousarray(x, dtype=np.float32), int(rate)


def _resample(x: np.ndarray, src: int, dst: int) -> np.ndarray:
    if src == dst or x.size == 0:
        return x
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(src, dst)
    return resample_poly(x.astype(np.float64), dst // g, src // g).astype(np.float32)


class AudioProcessor:
    """Static helpers, same names and semantics as the reference class."""

    @staticmethod
    def _read(path_or_bytes: Union[str, bytes]) -> Tuple[np.ndarray, int]:
        if isinstance(path_or_bytes, str):
            if not Path(path_or_bytes).exists():
                raise FileNotFoundError(f"Audio file not found: {path_or_bytes}")
            with open(path_or_bytes, "rb") as fh:
                return _parse_wav(fh.read())
        return _parse_wav(bytes(path_or_bytes))

    @staticmethod
    def probe_duration(path_or_bytes: Union[str, bytes]) -> float:
        x, rate = AudioProcessor._read(path_or_bytes)
        return len(x) / float(rate)

    @staticmethod
    def load_audio(path_or_bytes: Union[str, bytes], sample_rate: int) -> np.ndarray:
        x, rate = AudioProcessor._read(path_or_bytes)
        return AudioProcessor.normalize_to_int16(_resample(x, rate, sample_rate))

    @staticmethod
    def normalize_to_int16(audio: np.ndarray) -> np.ndarray:
        centred = audio - np.mean(audio)
        peak = np.max(np.abs(centred))
        if peak > 0:
            centred = centred * (29491.0 / peak)        # 90 % of full scale
        return centred.astype(np.int16)

    @staticmethod
    def fix_clipped_audio(audio: np.ndarray) -> np.ndarray:
        audio = np.nan_to_num(audio, nan=0.0, posinf=0.0, neginf=0.0)
        peak = np.max(np.abs(audio))
        if peak >= 32767:
            return (audio * (26214.0 / peak)).astype(np.int16)   # 80 % of full scale
        return audio

    @staticmethod
    def save_audio(audio: np.ndarray, file_path: str, sample_rate: int) -> None:
        if audio.size == 0:
            raise ValueError("Cannot save empty audio.")
        Path(file_path).parent.mkdir(parents=True, exist_ok=True)
        flat = np.asarray(audio).reshape(-1)
        if flat.dtype != np.int16:
            if np.issubdtype(flat.dtype, np.floating):
                flat = np.clip(flat * 32768.0 if np.max(np.abs(flat)) <= 1.0 else flat, -32768, 32767)
            flat = flat.astype(np.int16)
        payload = flat.astype("<i2").tobytes()
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, 1, sample_rate, sample_rate * 2, 2, 16, 22, 16, 0x4) + _PCM_GUID
        with open(file_path, "wb") as fh:
            fh.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload)) + b"WAVE")
            fh.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
            fh.write(b"data" + struct.pack("<I", len(payload)) + payload)

    @staticmethod
    def to_wav_bytes(audio: np.ndarray, sample_rate: int) -> bytes:
        buf = io.BytesIO()
        flat = np.asarray(audio).reshape(-1).astype("<i2")
        buf.write(b"RIFF" + struct.pack("<I", 36 + flat.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        buf.write(b"data" + struct.pack("<I", flat.nbytes) + flat.tobytes())
        return buf.getvalue()

    # ------------------------------------------------------------------ chunk joining
    @staticmethod
    def concatenate_with_crossfade(generated_waves: List[np.ndarray], cross_fade_duration: float, sample_rate: int) -> np.ndarray:
        if not generated_waves:
            return np.array([])
        if len(generated_waves) == 1:
            return generated_waves[0].reshape(-1)
        flat = [w.reshape(-1) for w in generated_waves]
        if cross_fade_duration <= 0:
            return np.concatenate(flat)
        out = flat[0]
        for nxt in flat[1:]:
            n = min(int(cross_fade_duration * sample_rate), len(out), len(nxt))
            if n <= 0:
                out = np.concatenate([out, nxt])
                continue
            ramp_down, ramp_up = np.linspace(1, 0, n), np.linspace(0, 1, n)
            out = np.concatenate([out[:-n], out[-n:] * ramp_down + nxt[:n] * ramp_up, nxt[n:]])
        return out

    @staticmethod
    def concatenate_with_crossfade_improved(generated_waves: List[np.ndarray], cross_fade_duration: float, sample_rate: int) -> np.ndarray:
        if not generated_waves:
            return np.array([])
        if len(generated_waves) == 1:
            return generated_waves[0].reshape(-1)
        flat = [AudioProcessor.fix_clipped_audio(w.reshape(-1)) for w in generated_waves]
        if cross_fade_duration <= 0:
            return np.concatenate(flat)
        out = flat[0]
        for nxt in flat[1:]:
            n = min(int(cross_fade_duration * sample_rate), len(out), len(nxt))
            if n <= 0:
                out = np.concatenate([out, nxt])
                continue
            tail, head = out[-n:], nxt[:n]
            rms_prev = np.sqrt(np.mean(tail.astype(np.float32) ** 2))
            rms_next = np.sqrt(np.mean(head.astype(np.float32) ** 2))
            if rms_prev > 100 and rms_next > 100:
                gain = np.clip(rms_prev / rms_next, 0.7, 1.5)          # level-match, bounded
                nxt = (nxt.astype(np.float32) * gain).astype(np.int16)
                head = nxt[:n]
            theta = np.linspace(0, np.pi / 2, n)
            mixed = (tail.astype(np.float32) * np.cos(theta) ** 2 + head.astype(np.float32) * np.sin(theta) ** 2).astype(np.int16)
            out = np.concatenate([out[:-n], mixed, nxt[n:]])
        return out


class CrossfadeStream:
    """Incremental form of ``concatenate_with_crossfade_improved`` for streaming output.

    The buffered join (reference core/audio_processor.py:122-192) repairs clipping once per RAW chunk and then, per
    junction, rewrites only the last ``n = min(cross_fade, len(out), len(next))`` samples of what has been joined.
    This class keeps exactly that state -- the not-yet-final tail of ``out`` and its total length -- so every chunk goes
    through ``fix_clipped_audio`` once, already-emitted samples are never touched again, and the concatenation of the
    blocks returned by ``push`` equals the buffered result sample for sample.  ``n_chunks`` is needed up front because
    the buffered function returns a single chunk untouched (no clip repair when there is nothing to join)."""

    def __init__(self, n_chunks: int, cross_fade_duration: float, sample_rate: int):
        self.n_chunks = int(n_chunks)
        self.cf = int(cross_fade_duration * sample_rate) if cross_fade_duration > 0 else 0
        self.fade = cross_fade_duration > 0
        self.held = None          # un-emitted tail of the joined signal
        self.total = 0            # samples joined so far (emitted + held)
        self.seen = 0

    def push(self, wave: np.ndarray) -> np.ndarray:
        """Add the next raw chunk; returns the samples that became final (possibly empty)."""
        self.seen += 1
        last = self.seen >= self.n_chunks
        w = np.asarray(wave).reshape(-1)
        if self.n_chunks == 1:
            return w
        nxt = AudioProcessor.fix_clipped_audio(w)
        if self.held is None:
            joined = nxt
        else:
            n = min(self.cf, self.total, len(nxt)) if self.fade else 0
            if n <= 0:
                joined = np.concatenate([self.held, nxt])
            else:
                tail, head = self.held[-n:], nxt[:n]
                rms_prev = np.sqrt(np.mean(tail.astype(np.float32) ** 2))
                rms_next = np.sqrt(np.mean(head.astype(np.float32) ** 2))
                if rms_prev > 100 and rms_next > 100:
                    gain = np.clip(rms_prev / rms_next, 0.7, 1.5)
                    nxt = (nxt.astype(np.float32) * gain).astype(np.int16)
                    head = nxt[:n]
                theta = np.linspace(0, np.pi / 2, n)
                mixed = (tail.astype(np.float32) * np.cos(theta) ** 2 + head.astype(np.float32) * np.sin(theta) ** 2).astype(np.int16)
                joined = np.concatenate([self.held[:-n], mixed, nxt[n:]])
        self.total += len(joined) - (0 if self.held is None else len(self.held))
        keep = 0 if last else min(self.cf, len(joined))       # the next junction may rewrite at most the last cf samples
        out, self.held = joined[: len(joined) - keep], joined[len(joined) - keep:]
        return np.ascontiguousarray(out)

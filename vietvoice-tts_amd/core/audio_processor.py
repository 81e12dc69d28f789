"""AudioProcessor -- host-side audio plumbing of the hot path, behaviour-identical to the reference
(vietvoicetts/core/audio_processor.py) for every pure-numpy step: DC removal + peak 29491 + int16
truncation (:28-44), clip repair (:46-58), linear cross-fade (:69-120), RMS-matched cos^2 cross-fade
(:122-192).  Pinned by tests/golden/host_golden.npz (generated from the reference functions).

``load_audio`` (:15-26) is ``AudioSegment.from_file(..).set_channels(1).set_frame_rate(sr)`` followed by
``get_array_of_samples`` -> float32 -> ``normalize_to_int16``.  pydub (pin: >=0.25.0, pyproject.toml:38) is absent
offline; for RIFF/WAVE PCM input its two calls are integer arithmetic of CPython's stdlib ``audioop``
(pydub/audio_segment.py ``set_channels``: ``audioop.tomono(data, width, 0.5, 0.5)``; ``set_frame_rate``:
``audioop.ratecv(data, width, channels, src, dst, None)``), restated here in closed form with numpy:

  tomono : floor(l * 0.5 + r * 0.5)                                    = (l + r) >> 1
  ratecv : output m sits at input position m * I / O (I, O = rates / gcd); with n1 = ceil(m * I / O),
           d = n1 * O - m * I in [0, O):  out = trunc((x32[n1 - 1] * d + x32[n1] * (O - d)) / O) >> (32 - 8 * width)
           (x32 = sample << (32 - 8 * width), x32[-1] = 0; products, sum and the truncated quotient are exact in
           float64 for the 16-bit case and O < 65536, where the whole expression equals floor((p * d + c * (O - d)) / O));
           floor((N - 1) * O / I) + 1 outputs for N input frames.

Bit-exact against stdlib ``audioop`` (tests/test_ingest_cpu.py, tests/golden/ingest_golden.npz) for widths 1, 2, 4.
pydub's own glue is restated from its published source and is "parity unpinned": 8-bit WAV is unsigned and gets
``audioop.bias(-128)``; 24-bit samples are widened to 32 bit with the sign byte written FIRST (value * 256 + 0x00 / 0xFF);
more than two channels are mixed as ``sum(sample // channels)``.  Float WAVE goes through ffmpeg in the reference
(``-acodec pcm_s32le``): restated as ``clip(llrint(x * 2**31))`` and unpinned; other containers need ffmpeg and are
out of scope.  ``save_audio`` writes a WAVE_FORMAT_EXTENSIBLE PCM16 file by hand (soundfile 'WAVEX' bytes unpinned).
The polyphase resampler of earlier rounds stays available as an explicit opt-in (``resampler="polyphase"``).
"""
from __future__ import annotations

import io
import struct
from math import gcd
from pathlib import Path
from typing import List, Tuple, Union

import numpy as np

_PCM_GUID = bytes.fromhex("0100000000001000800000aa00389b71")
_INT_OF_WIDTH = {1: np.int8, 2: np.int16, 4: np.int32}


def _decode_wav(data: bytes) -> Tuple[np.ndarray, int, int]:
    """RIFF/WAVE bytes -> (frames (n_frames, channels) of int8 / int16 / int32, sample width in bytes, rate): the
    samples pydub's AudioSegment holds after ``from_file`` (see the module docstring for the 8- / 24-bit / float glue)."""
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
    if fmt is None or pcm is None or len(fmt) < 16:
        raise ValueError("malformed WAVE file: missing fmt or data chunk")
    tag, ch, rate, _br, _ba, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack("<H", fmt[24:26])[0]
    if ch < 1 or rate < 1:
        raise ValueError("malformed WAVE file: zero channels or rate")
    if tag == 1:
        if bits == 8:
            x, width = (np.frombuffer(pcm, dtype=np.uint8) ^ 0x80).view(np.int8), 1          # audioop.bias(data, 1, -128)
        elif bits == 16:
            x, width = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2"), 2
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3)
            w = np.empty((b.shape[0], 4), np.uint8)
            w[:, 0] = np.where(b[:, 2] > 0x7F, 0xFF, 0x00)                                     # pydub writes the pad byte first
            w[:, 1:] = b
            x, width = w.reshape(-1).view("<i4"), 4
        elif bits == 32:
            x, width = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4"), 4
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3 and bits in (32, 64):
        f = np.frombuffer(pcm[: len(pcm) // (bits // 8) * (bits // 8)], dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
        x, width = np.clip(np.rint(np.nan_to_num(f) * 2147483648.0), -2147483648.0, 2147483647.0).astype(np.int32), 4
    else:
        raise ValueError(f"unsupported WAVE format tag {tag}")
    x = x.astype(_INT_OF_WIDTH[width], copy=False)
    return np.ascontiguousarray(x[: len(x) // ch * ch].reshape(-1, ch)), width, int(rate)


def tomono(frames: np.ndarray) -> np.ndarray:
    """(n, channels) integer frames -> (n,) mono, pydub ``set_channels(1)``: 2 channels = audioop.tomono(.., 0.5, 0.5)
    (floor of the half sum), more = sum of floor-divided samples (OverflowError like pydub's array arithmetic when the
    floors add up below the range: only with every channel at negative full scale)."""
    ch = frames.shape[1]
    if ch == 1:
        return frames[:, 0]
    if ch == 2:
        return ((frames[:, 0].astype(np.int64) + frames[:, 1].astype(np.int64)) >> 1).astype(frames.dtype)
    mixed = (frames.astype(np.int64) // ch).sum(axis=1)
    info = np.iinfo(frames.dtype)
    if mixed.size and (mixed.min() < info.min or mixed.max() > info.max):     # pydub accumulates in an array.array of the width
        raise OverflowError("channel mix does not fit the sample width (every channel at negative full scale)")
    return mixed.astype(frames.dtype)


def ratecv_len(n_in: int, src: int, dst: int) -> int:
    """Frames audioop.ratecv(.., state=None) emits for n_in input frames."""
    if n_in <= 0:
        return 0
    g = gcd(src, dst)
    return (n_in - 1) * (dst // g) // (src // g) + 1


def ratecv(x: np.ndarray, src: int, dst: int) -> np.ndarray:
    """Mono integer samples at ``src`` Hz -> ``dst`` Hz, audioop.ratecv(data, width, 1, src, dst, None) (weights 1, 0):
    linear interpolation between the two neighbouring input samples, evaluated in float64 on the samples shifted to
    32 bit exactly as Modules/audioop.c does, truncated, shifted back."""
    if src == dst or x.size == 0:                      # pydub returns the segment itself / skips empty data
        return x
    g = gcd(src, dst)
    I, O = src // g, dst // g
    shift = 32 - 8 * x.dtype.itemsize
    m = np.arange(ratecv_len(x.size, src, dst), dtype=np.int64)
    n1 = (m * I + O - 1) // O
    d = (n1 * O - m * I).astype(np.float64)
    x32 = (x.astype(np.int64) << shift).astype(np.float64)
    prev = np.where(n1 > 0, x32[np.maximum(n1 - 1, 0)], 0.0)
    out = ((prev * d + x32[n1] * (float(O) - d)) / float(O)).astype(np.int64)        # C (int) cast: truncation
    return (out >> shift).astype(x.dtype)


def _resample_polyphase(x: np.ndarray, src: int, dst: int) -> np.ndarray:
    """Opt-in band-limited resampler (NOT the reference's arithmetic)."""
    if src == dst or x.size == 0:
        return x
    from scipy.signal import resample_poly
    g = gcd(src, dst)
    return resample_poly(x.astype(np.float64), dst // g, src // g).astype(np.float32)


class AudioProcessor:
    """Static helpers, same names and semantics as the reference class."""

    @staticmethod
    def _read_bytes(path_or_bytes: Union[str, bytes]) -> bytes:
        if isinstance(path_or_bytes, str):
            if not Path(path_or_bytes).exists():
                raise FileNotFoundError(f"Audio file not found: {path_or_bytes}")
            with open(path_or_bytes, "rb") as fh:
                return fh.read()
        return bytes(path_or_bytes)

    @staticmethod
    def decode(path_or_bytes: Union[str, bytes]) -> Tuple[np.ndarray, int, int]:
        """-> (integer frames (n, channels), sample width, rate); container parsing only, no arithmetic on samples."""
        return _decode_wav(AudioProcessor._read_bytes(path_or_bytes))

    @staticmethod
    def probe_duration(path_or_bytes: Union[str, bytes]) -> float:
        frames, _w, rate = AudioProcessor.decode(path_or_bytes)
        return frames.shape[0] / float(rate)

    @staticmethod
    def load_samples(path_or_bytes: Union[str, bytes], sample_rate: int) -> np.ndarray:
        """The integer samples ``audio_segment.get_array_of_samples()`` holds in the reference (:22-25): mono, at sample_rate."""
        frames, _w, rate = AudioProcessor.decode(path_or_bytes)
        return ratecv(tomono(frames), rate, sample_rate)

    @staticmethod
    def load_audio(path_or_bytes: Union[str, bytes], sample_rate: int, resampler: str = "ratecv") -> np.ndarray:
        if resampler == "ratecv":
            return AudioProcessor.normalize_to_int16(AudioProcessor.load_samples(path_or_bytes, sample_rate).astype(np.float32))
        if resampler != "polyphase":
            raise ValueError(f"unknown resampler {resampler!r}")
        frames, _w, rate = AudioProcessor.decode(path_or_bytes)
        x = frames.astype(np.float32).mean(axis=1)              # the normalisation below is scale-free
        return AudioProcessor.normalize_to_int16(_resample_polyphase(x, rate, sample_rate))

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

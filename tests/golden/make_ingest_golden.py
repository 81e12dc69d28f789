"""Generates tests/golden/ingest_golden.json + ingest_golden.npz: what the REFERENCE's AudioProcessor.load_audio
(vietvoicetts/core/audio_processor.py:15-26) returns for RIFF/WAVE clips, with pydub's two calls supplied by CPython's
stdlib ``audioop`` -- the library pydub itself calls.

Run in the build container only:   python tests/golden/make_ingest_golden.py

pydub (pin >=0.25.0, pyproject.toml:38) is absent offline.  Its published AudioSegment does, for WAV input:
  from_file            -> sample bytes of the data chunk, width / channels / rate of the fmt chunk (stdlib ``wave`` here);
                          8-bit WAV is unsigned: audioop.bias(data, 1, -128)
  set_channels(1)      -> audioop.tomono(data, width, 0.5, 0.5) for stereo
  set_frame_rate(sr)   -> audioop.ratecv(data, width, channels, rate, sr, None)[0] when the rate differs and data is not empty
  get_array_of_samples -> array.array(typecode of the width, data)
The class below is that glue over stdlib audioop and stands where ``from pydub import AudioSegment`` would; the reference
module is loaded BY FILE PATH (same technique as make_host_golden.py) and its own load_audio / normalize_to_int16 run
unmodified on top of it.  Stored per case: the generator parameters of the input clip (seeded; plus the sha1 of the WAV
bytes so RNG drift is detected), ``samples`` = get_array_of_samples() (pure audioop) and ``pcm`` = load_audio's int16 result.
Widths 1 and 4 are recorded for the audioop steps too; pydub's glue for them (and for 24-bit) is restated from memory of
its source and stays "parity unpinned".  No reference source text is stored.
"""
import array
import hashlib
import io
import json
import os
import struct
import sys
import wave

import numpy as np

try:                                  # stdlib up to Python 3.12; only the generator itself needs it
    import audioop
except ImportError:                   # pragma: no cover
    audioop = None

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.dirname(os.path.abspath(__file__))

TYPECODE = {1: "b", 2: "h", 4: "i"}


class AudioSegment:
    """pydub.AudioSegment for WAV input, restricted to the four calls the reference makes."""

    def __init__(self, data, width, channels, rate):
        self._data, self.sample_width, self.channels, self.frame_rate = data, width, channels, rate

    @classmethod
    def from_file(cls, fh):
        with wave.open(fh, "rb") as w:
            width, ch, rate = w.getsampwidth(), w.getnchannels(), w.getframerate()
            data = w.readframes(w.getnframes())
        if width == 1:
            data = audioop.bias(data, 1, -128)
        return cls(data, width, ch, rate)

    def set_channels(self, channels):
        if channels == self.channels:
            return self
        assert channels == 1 and self.channels == 2
        return AudioSegment(audioop.tomono(self._data, self.sample_width, 0.5, 0.5), self.sample_width, 1, self.frame_rate)

    def set_frame_rate(self, frame_rate):
        if frame_rate == self.frame_rate:
            return self
        data = audioop.ratecv(self._data, self.sample_width, self.channels, self.frame_rate, frame_rate, None)[0] if self._data else self._data
        return AudioSegment(data, self.sample_width, self.channels, frame_rate)

    def get_array_of_samples(self):
        return array.array(TYPECODE[self.sample_width], self._data)


def make_clip(seed: int, n_frames: int, channels: int, rate: int, width: int) -> np.ndarray:
    """Seeded speech-band test signal with a DC offset and full-scale excursions: (n_frames, channels) integers of `width`."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_frames)[:, None] / float(rate)
    f = rng.uniform(80.0, min(7600.0, 0.45 * rate), size=(1, 12))
    sig = np.zeros((n_frames, channels))
    for c in range(channels):
        ph = rng.uniform(0, 2 * np.pi, size=(1, 12))
        amp = rng.uniform(0.2, 1.0, size=(1, 12))
        sig[:, c] = (amp * np.sin(2 * np.pi * f * t + ph)).sum(axis=1) + 0.3 * rng.standard_normal(n_frames) + 0.15 * (c + 1)
    full = float(1 << (8 * width - 1))
    x = np.clip(np.rint(sig / np.abs(sig).max() * 0.98 * full), -full, full - 1).astype({1: np.int8, 2: np.int16, 4: np.int32}[width])
    if n_frames >= 4:                                      # exact extremes next to each other: rounding at the range ends
        x[0], x[1], x[2] = -int(full), int(full) - 1, -int(full)
        if channels == 2:
            x[1, 1], x[2, 1] = -int(full), int(full) - 1
    return x


def wav_bytes(frames: np.ndarray, rate: int, width: int) -> bytes:
    raw = frames.astype({1: np.int8, 2: "<i2", 4: "<i4"}[width])
    payload = (raw.view(np.uint8) ^ 0x80).tobytes() if width == 1 else raw.tobytes()       # 8-bit WAV is unsigned
    ch = frames.shape[1]
    hdr = struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * width, ch * width, 8 * width)
    return b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + hdr + b"data" + struct.pack("<I", len(payload)) + payload


# (seed, n_frames, channels, rate, width): 16-bit mono + stereo at the rates the verdict names, 24 kHz pass-through,
# odd rates (O up to 24000), tiny clips, one clip long enough for several 8192-element numpy buffers; widths 1 and 4
CASES = [(100 + i, n, ch, rate, 2) for i, (n, ch, rate) in enumerate([
    (2000, 1, 8000), (2000, 2, 8000), (4000, 1, 16000), (4000, 2, 16000), (5513, 1, 22050), (5513, 2, 22050),
    (11025, 1, 44100), (11025, 2, 44100), (12000, 1, 48000), (12000, 2, 48000), (6000, 1, 24000), (6000, 2, 24000),
    (3000, 1, 11025), (3001, 2, 32000), (2500, 1, 44101), (2500, 2, 23999), (1, 1, 48000), (2, 2, 8000), (3, 1, 16000),
    (150000, 2, 44100), (70000, 1, 24000)])]
CASES += [(200, 3000, 2, 16000, 1), (201, 3000, 1, 44100, 1), (202, 3000, 2, 48000, 4), (203, 3000, 1, 22050, 4)]


def main():
    import types
    from make_host_golden import load_reference
    pd = types.ModuleType("pydub")
    pd.AudioSegment = AudioSegment
    sys.modules["pydub"] = pd                              # before the reference module binds `from pydub import AudioSegment`
    ref_ap = load_reference()["audio_processor"].AudioProcessor
    gold, arrays = [], {}
    for seed, n, ch, rate, width in CASES:
        frames = make_clip(seed, n, ch, rate, width)
        wav = wav_bytes(frames, rate, width)
        seg = AudioSegment.from_file(io.BytesIO(wav)).set_channels(1).set_frame_rate(24000)
        samples = np.array(seg.get_array_of_samples())
        pcm = ref_ap.load_audio(wav, 24000)                # the reference's own function, bytes branch (:24-26)
        assert pcm.dtype == np.int16 and pcm.shape == samples.shape
        key = f"s{seed}"
        arrays[key + "_samples"], arrays[key + "_pcm"] = samples, pcm
        gold.append(dict(key=key, seed=seed, n_frames=n, channels=ch, rate=rate, width=width, n_out=int(samples.size),
                         wav_sha1=hashlib.sha1(wav).hexdigest(), pinned_glue=(width == 2)))
    with open(os.path.join(OUT, "ingest_golden.json"), "w") as f:
        json.dump(dict(dst_rate=24000, numpy=np.__version__, python=sys.version.split()[0], cases=gold), f, indent=1)
    np.savez_compressed(os.path.join(OUT, "ingest_golden.npz"), **arrays)
    print("wrote", len(gold), "cases")


if __name__ == "__main__":
    main()

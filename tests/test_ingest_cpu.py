"""Reference-clip ingest (a8 / N3), CPU side: the host loader and the loop oracle against

  * tests/golden/ingest_golden.npz -- the REFERENCE's AudioProcessor.load_audio run over stdlib audioop
    (tests/golden/make_ingest_golden.py; reference core/audio_processor.py:15-26), and
  * stdlib ``audioop`` / ``numpy.mean`` run live on random clips (audioop is stdlib up to Python 3.12; skipped without it).

Bit-exact everywhere: this is integer work.  Widths 1 and 4 pin the audioop arithmetic only; pydub's glue around them
(8-bit bias, 24-bit widening, > 2 channels) is restated from its published source and is "parity unpinned".
"""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ingest_golden.json")))
NPZ = np.load(os.path.join(HERE, "golden", "ingest_golden.npz"))


def _gen():
    spec = importlib.util.spec_from_file_location("make_ingest_golden", os.path.join(HERE, "golden", "make_ingest_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GEN = _gen()


def case_wav(c):
    frames = GEN.make_clip(c["seed"], c["n_frames"], c["channels"], c["rate"], c["width"])
    wav = GEN.wav_bytes(frames, c["rate"], c["width"])
    assert hashlib.sha1(wav).hexdigest() == c["wav_sha1"], "seeded input clip is not the one the fixture was made from"
    return frames, wav


@pytest.mark.parametrize("c", GOLD["cases"], ids=lambda c: f"{c['key']}-{c['channels']}ch-{c['rate']}-w{c['width']}")
def test_host_loader_matches_reference_fixture(c, tmp_path):
    from vietvoice_tts_amd.core import AudioProcessor
    from vietvoice_tts_amd.core.audio_processor import ratecv_len
    _frames, wav = case_wav(c)
    samples = AudioProcessor.load_samples(wav, GOLD["dst_rate"])
    assert samples.dtype == NPZ[c["key"] + "_samples"].dtype and np.array_equal(samples, NPZ[c["key"] + "_samples"])
    assert c["n_out"] == (ratecv_len(c["n_frames"], c["rate"], 24000) if c["rate"] != 24000 else c["n_frames"])
    pcm = AudioProcessor.load_audio(wav, GOLD["dst_rate"])
    assert pcm.dtype == np.int16 and np.array_equal(pcm, NPZ[c["key"] + "_pcm"])
    p = tmp_path / "clip.wav"                                  # the path branch (:18-22) reads the same bytes
    p.write_bytes(wav)
    assert np.array_equal(AudioProcessor.load_audio(str(p), GOLD["dst_rate"]), pcm)


def test_loop_oracle_matches_fixture_and_closed_form():
    """oracle/ingest_oracle.py (audioop.c's counter-based state machine, literally) == fixture == the product's closed form."""
    from oracle import ingest_oracle as io_
    from vietvoice_tts_amd.core.audio_processor import ratecv, tomono
    for c in GOLD["cases"]:
        if c["n_frames"] > 12000:
            continue                                           # pure-Python loops: small cases only
        frames, _wav = case_wav(c)
        mono = frames[:, 0].tolist() if c["channels"] == 1 else io_.tomono(frames[:, 0].tolist(), frames[:, 1].tolist(), c["width"])
        assert mono == tomono(frames).tolist()
        out = io_.ratecv(mono, c["width"], c["rate"], 24000) if c["rate"] != 24000 else mono
        assert out == NPZ[c["key"] + "_samples"].tolist(), c["key"]
        assert out == ratecv(tomono(frames), c["rate"], 24000).tolist()


def test_live_audioop_random_clips():
    audioop = pytest.importorskip("audioop")
    from vietvoice_tts_amd.core.audio_processor import ratecv, ratecv_len, tomono
    rng = np.random.default_rng(77)
    for width, dt in ((1, np.int8), (2, np.int16), (4, np.int32)):
        info = np.iinfo(dt)
        for src in (8000, 16000, 22050, 44100, 48000, 11025, 32000, 44101, 96000, 23999, 24001, 12345, 7, 1000000):
            for n in (1, 2, 3, 17, 1000, 4801):
                x = rng.integers(info.min, info.max + 1, size=(n, 2)).astype(dt)
                if n > 3:
                    x[0], x[1], x[2] = info.min, (info.min, info.max), info.max
                want_mono = np.frombuffer(audioop.tomono(x.tobytes(), width, 0.5, 0.5), dtype=dt)
                assert np.array_equal(tomono(x), want_mono)
                want = np.frombuffer(audioop.ratecv(want_mono.tobytes(), width, 1, src, 24000, None)[0], dtype=dt)
                got = ratecv(want_mono, src, 24000)
                assert got.dtype == dt and np.array_equal(got, want), (width, src, n)
                assert ratecv_len(n, src, 24000) == len(want)


def test_float32_mean_order_is_numpys():
    """The summation order the GPU kernel reproduces (oracle/ingest_oracle.py::float32_mean) IS numpy's, at lengths around
    every boundary of the scheme (8, 128-leaf, the (n/2) & ~7 split, the 8192 buffer)."""
    from oracle.ingest_oracle import float32_mean
    rng = np.random.default_rng(5)
    for n in (1, 5, 7, 8, 9, 127, 128, 129, 130, 143, 144, 255, 257, 1000, 4097, 8191, 8192, 8193, 16384, 16385, 24000, 65537):
        for scale, dc in ((3000.0, 100.0), (1.0, 0.0), (9000.0, -4000.0)):
            a = (rng.standard_normal(n) * scale + dc).astype(np.float32)
            assert float32_mean(a) == np.mean(a), (n, scale)


def test_more_than_two_channels_and_odd_containers():
    """pydub's > 2-channel mix (sum of floor-divided samples), 24-bit widening and 8-bit bias as restated; unpinned glue."""
    from vietvoice_tts_amd.core import AudioProcessor
    from vietvoice_tts_amd.core.audio_processor import tomono, _decode_wav
    import struct
    x = np.array([[-7, 5, 3], [32767, 32767, 32767], [-32768, -32768, -32766], [-1, -1, -1]], np.int16)
    assert tomono(x).tolist() == [(-7 // 3) + (5 // 3) + (3 // 3), 32766, -10923 - 10923 - 10922, -3]
    with pytest.raises(OverflowError):                         # -10923 * 3 = -32769: array.array('h') arithmetic overflows in pydub
        tomono(np.full((2, 3), -32768, np.int16))
    # 24-bit: bytes b0 b1 b2 -> int32 with the sign byte written first (value * 256 + 0x00 / 0xFF)
    vals = [0, 1, -1, 8388607, -8388608, 12345, -54321]
    payload = b"".join(struct.pack("<i", v)[:3] for v in vals)
    hdr = struct.pack("<IHHIIHH", 16, 1, 1, 24000, 72000, 3, 24)
    wav = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + hdr + b"data" + struct.pack("<I", len(payload)) + payload
    frames, width, rate = _decode_wav(wav)
    assert width == 4 and rate == 24000 and frames[:, 0].tolist() == [v * 256 + (255 if v < 0 else 0) for v in vals]
    # 8-bit: unsigned in the file, signed after the bias
    payload = bytes([0, 128, 255, 127])
    hdr = struct.pack("<IHHIIHH", 16, 1, 1, 8000, 8000, 1, 8)
    wav = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + hdr + b"data" + struct.pack("<I", len(payload)) + payload
    frames, width, _ = _decode_wav(wav)
    assert width == 1 and frames[:, 0].tolist() == [-128, 0, 127, -1]
    with pytest.raises(ValueError):
        AudioProcessor.load_audio(b"RIFF\x04\x00\x00\x00WAVE", 24000)
    assert AudioProcessor.load_audio(GEN.wav_bytes(GEN.make_clip(1, 4800, 2, 48000, 2), 48000, 2), 24000, resampler="polyphase").dtype == np.int16

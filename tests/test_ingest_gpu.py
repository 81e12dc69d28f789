"""-m gpu: reference-clip ingest on the device (a8 / N3), through the C ABI, BIT-EXACT against

  * tests/golden/ingest_golden.npz -- the REFERENCE's AudioProcessor.load_audio run over stdlib audioop
    (reference core/audio_processor.py:15-26; generator tests/golden/make_ingest_golden.py), and
  * numpy's normalize arithmetic (np.mean's float32 summation order included) on lengths around every boundary of
    numpy's pairwise scheme.
Integer work: the bar is equality, sample for sample."""
import numpy as np
import pytest

from tests.test_ingest_cpu import GOLD, NPZ, GEN, case_wav

pytestmark = pytest.mark.gpu


def _bank(eng, **kw):
    from vietvoice_tts_amd.voice_bank import VoiceBank
    return VoiceBank(eng, 24000, **kw)


def test_ingest_kernel_samples_equal_audioop(hip_tiny):
    """vv_ingest_pcm alone: float32 of get_array_of_samples() for every fixture case, all clips in ONE launch."""
    import torch
    from vietvoice_tts_amd.core import AudioProcessor
    eng = hip_tiny["f32"]
    bank = _bank(eng)
    decoded = [AudioProcessor.decode(case_wav(c)[1]) for c in GOLD["cases"]]
    x, lens = bank._ingest_ratecv(decoded, eng.device)
    torch.cuda.synchronize()
    x = x.cpu().numpy()
    assert lens == [c["n_out"] for c in GOLD["cases"]]
    off = np.concatenate([[0], np.cumsum(lens)])
    for j, c in enumerate(GOLD["cases"]):
        want = NPZ[c["key"] + "_samples"].astype(np.float32)             # np.array(samples, dtype=np.float32) (:25)
        assert np.array_equal(x[off[j]: off[j + 1]], want), c["key"]


def test_voice_bank_pcm_equals_reference_load_audio(hip_tiny, tmp_path):
    """The whole device path (ingest + numpy-ordered mean + peak + scale + truncation) == the reference's load_audio, bit for bit,
    batched and one by one, bytes and path inputs."""
    eng = hip_tiny["f32"]
    wavs = [case_wav(c)[1] for c in GOLD["cases"]]
    ents = _bank(eng).ingest_many(wavs)
    for c, e in zip(GOLD["cases"], ents):
        want = NPZ[c["key"] + "_pcm"]
        assert e.pcm_host.dtype == np.int16 and np.array_equal(e.pcm_host, want), c["key"]
        assert np.array_equal(e.pcm_dev.cpu().numpy(), want)
    one = _bank(eng)
    for c, w in list(zip(GOLD["cases"], wavs))[:6]:
        p = tmp_path / (c["key"] + ".wav")
        p.write_bytes(w)
        assert np.array_equal(one.get(str(p)).pcm_host, NPZ[c["key"] + "_pcm"])


@pytest.mark.parametrize("n", [1, 5, 7, 8, 9, 127, 128, 129, 130, 143, 144, 255, 257, 1000, 4097, 8191, 8192, 8193, 16384, 16385,
                               24000, 65537, 191999, 700001])
def test_normalize_clips_bit_exact_numpy(hip_tiny, n):
    """vv_normalize_clips == the reference's normalize_to_int16 (host mirror pinned by reference-generated vectors in
    test_host_cpu.py) bit for bit: the mean is summed in numpy's order, so no sample truncates differently."""
    import torch
    from vietvoice_tts_amd.core import AudioProcessor
    eng = hip_tiny["f32"]
    rng = np.random.default_rng(n)
    clips = [(rng.standard_normal(n) * s + o).astype(np.float32) for s, o in ((3000.0, 120.0), (0.1, 0.0), (9000.0, -400.0))]
    clips.append(np.rint(rng.standard_normal(n) * 8000 + 500).astype(np.float32))           # integer-valued, like real samples
    off = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    out = eng.normalize_clips(torch.from_numpy(np.concatenate(clips)).cuda(), torch.from_numpy(off).cuda(), max_len=n).cpu().numpy()
    for j, c in enumerate(clips):
        assert np.array_equal(out[off[j]: off[j + 1]], AudioProcessor.normalize_to_int16(c)), (n, j)


def test_normalize_clips_degenerate(hip_tiny):
    import torch
    from vietvoice_tts_amd.core import AudioProcessor
    eng = hip_tiny["f32"]
    clips = [np.zeros(500, np.float32), np.full(300, 7.0, np.float32), np.array([5.0], np.float32)]     # silence, pure DC, one sample
    off = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    out = eng.normalize_clips(torch.from_numpy(np.concatenate(clips)).cuda(), torch.from_numpy(off).cuda()).cpu().numpy()
    for j, c in enumerate(clips):
        assert np.array_equal(out[off[j]: off[j + 1]], AudioProcessor.normalize_to_int16(c))


def test_many_channels_and_widths_on_device(hip_tiny):
    """> 2 channels (sum of floor-divided samples), 8- and 32-bit widths: device == host loader (glue unpinned, audioop pinned)."""
    from vietvoice_tts_amd.core import AudioProcessor
    eng = hip_tiny["f32"]
    rng = np.random.default_rng(3)
    wavs = []
    for ch, rate, width in ((3, 48000, 2), (6, 44100, 2), (4, 16000, 1), (5, 22050, 4), (3, 24000, 2)):
        info = np.iinfo({1: np.int8, 2: np.int16, 4: np.int32}[width])
        frames = rng.integers(info.min // 2, info.max // 2, size=(3001, ch)).astype(info.dtype)
        wavs.append(GEN.wav_bytes(frames, rate, width))
    for w, e in zip(wavs, _bank(eng).ingest_many(wavs)):
        assert np.array_equal(e.pcm_host, AudioProcessor.load_audio(w, 24000))


def test_polyphase_is_opt_in(hip_tiny):
    from vietvoice_tts_amd.core import AudioProcessor
    eng = hip_tiny["f32"]
    wav = case_wav(GOLD["cases"][9])[1]                                   # 48 kHz stereo
    got = _bank(eng, resampler="polyphase").get(wav).pcm_host
    want = AudioProcessor.load_audio(wav, 24000, resampler="polyphase")
    assert got.shape == want.shape and int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()) <= 1
    assert not np.array_equal(got, NPZ[GOLD["cases"][9]["key"] + "_pcm"])                 # and it is NOT the reference's arithmetic

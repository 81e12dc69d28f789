"""-m gpu: edge cases of the hot path -- the shapes the reference's own tests poke at on the host side (empty / minimal / ragged
inputs: /root/reference/tests/test_edge_cases.py:137-214, tests/test_text_processor.py:93-107) carried down to the kernels that
replace the three graphs: a minimal utterance (a 3-frame reference clip, 0 or 1 text ids, 1 generated frame), the conv_post stream
kernel's scalar / short-channel-block / ragged-length forms, and the ABI's argument errors (a code + message, never an abort)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,Cc,T,lens,ld_extra", [
    (2, 32, 3000, None, 0),            # the decode's form: T % 4 == 0, C % 8 == 0
    (2, 32, 3001, None, 3),            # rows not 16-byte aligned: four dword loads per lane, scalar PCM stores, ld_pcm != T
    (3, 12, 2996, [2996, 1777, 5], 0), # C % 8 != 0 -> 4-channel blocks; valid lengths not multiples of 4, one shorter than a window
    (2, 7, 1000, [999, 1], 0),         # C % 4 != 0 -> one channel per block
    (1, 64, 249 * 4, None, 0),         # one sample more than a wave's 248 outputs x 4 waves: the second workgroup holds one lane of work
    (1, 8, 4, [4], 0),                 # a single float4
])
def test_conv_post_stream_kernel_forms(hip_tiny, B, Cc, T, lens, ld_extra):
    """K13 against torch: waveform to fp32 tolerance, PCM +-1 LSB, samples past an item's valid input length see zeros (the conv's
    own padding rule), the halo lanes / wave seams / row ends exact."""
    from tests import gpu_util as gu
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(1000 + Cc + T)
    x = torch.randn(B, Cc, T, generator=g)
    w = torch.randn(1, Cc, 7, generator=g) / math.sqrt(Cc * 7) * 0.7
    bias = -0.03
    xm = x.clone()
    if lens is not None:
        for b, L in enumerate(lens):
            xm[b, :, L:] = 0.0                                  # what the kernel must see past the valid length
    ref = torch.tanh(F.conv1d(F.leaky_relu(xm, 0.01), w, torch.tensor([bias]), padding=3)).reshape(B, T)
    ld = T + ld_extra
    pcm = torch.full((B, ld), 12345, dtype=torch.int16, device=DEV)
    wave = torch.zeros(B, T, device=DEV)
    dx, dw = x.to(DEV), w.reshape(Cc, 7).contiguous().to(DEV)
    dl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=DEV)
    gu.check(eng, eng.lib.vv_conv_post(eng.ctx, dx.data_ptr(), dw.data_ptr(), bias, pcm.data_ptr(), ld, wave.data_ptr(), B, Cc, T, 7, 0.01,
                                       None if dl is None else dl.data_ptr(), gu.stream()))
    torch.cuda.synchronize()
    assert float((wave.cpu() - ref).abs().max()) < 2e-6
    ref_pcm = torch.clamp(ref * 32767.0, -32768.0, 32767.0).to(torch.int16)
    assert int((pcm[:, :T].cpu().int() - ref_pcm.int()).abs().max()) <= 1
    if ld_extra:
        assert bool((pcm[:, T:] == 12345).all()), "nothing is written past T"


@pytest.mark.parametrize("text_len", [1, 0])
def test_minimal_utterance_matches_oracle(hip_tiny, tiny_setup, text_len):
    """Near the smallest inputs the front end admits: a 600-sample reference clip (600 // 256 + 1 = 3 frames, core/tts_engine.py:55;
    the centred STFT's reflect padding needs more than n_fft / 2 = 512 samples, in the oracle's torch.stft as in any STFT), 1 or 0
    text ids, 1 generated frame (N = 4), next to a normal item in the same batch."""
    spec, w, orc = tiny_setup
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(5)
    la, lt, gf = [600, 256 * 9 + 17], [text_len, 13], [1, 7]
    B = 2
    audio = torch.zeros(B, max(la), dtype=torch.int16)
    ids = torch.zeros(B, max(max(lt), 1), dtype=torch.int32)
    for b in range(B):
        audio[b, : la[b]] = (torch.randn(la[b], generator=g) * 5000).clamp(-30000, 30000).to(torch.int16)
        if lt[b]:
            ids[b, : lt[b]] = torch.randint(1, spec.vocab_size, (lt[b],), generator=g, dtype=torch.int32)
    seq = [la[b] // spec.hop_length + 1 + gf[b] for b in range(B)]
    N = max(seq)
    noise = torch.randn(B, N, spec.n_mel, generator=g)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    pre = eng.preprocess(audio.to(DEV), i32(la), ids.to(DEV), i32(lt), i32(seq), N)
    x = noise.to(DEV).clone()
    eng.transformer_steps(x, pre, 0, eng.n_steps)
    pcm, pcm_len = eng.decode_bucketed(x, pre, gf, min_units=1)
    torch.cuda.synchronize()
    for b in range(B):
        with torch.no_grad():
            p = orc.preprocess(audio[b, : la[b]], ids[b, : lt[b]], seq[b], noise[b, : seq[b]])
            xr = p["noise"]
            for st in range(orc.nfe_step - 1):
                xr = orc.transformer_step(xr, p, st)
            pr = orc.decode(xr, p["ref_signal_len"])
        assert int(pre["ref_signal_len"][b]) == p["ref_signal_len"] == la[b] // 256 + 1
        err = float((x[b, : seq[b]].cpu() - xr).abs().max() / xr.abs().max())
        n = pr.numel()
        assert int(pcm_len[b]) == n == gf[b] * spec.hop_length
        d = int((pcm[b, :n].cpu().int() - pr.int()).abs().max())
        print(f"\n[minimal, text_len {text_len}] item {b}: N = {seq[b]}, state rel err {err:.2e}, PCM max diff {d} LSB")
        assert err < 2e-4 and d <= 2


def test_abi_argument_errors_are_codes_not_aborts(hip_tiny, tiny_setup):
    """Every entry point returns 0 or a negative errno-style code and leaves a message (include/vvtts.h): lengths outside [1, N],
    a misaligned operand, an unsupported kernel width -- and the context keeps working afterwards."""
    from tests import gpu_util as gu
    from vietvoice_tts_amd import runtime as rt
    spec, w, _ = tiny_setup
    eng = hip_tiny["f32"]
    B, N = 2, 40
    x = torch.zeros(B, N, spec.n_mel, device=DEV)
    pre = {"cat_mel_text": torch.zeros(B, N, spec.cond_dim, device=DEV), "cat_mel_text_drop": torch.zeros(B, N, spec.cond_dim, device=DEV),
           "rope_cos_q": eng.rope[0][:N], "rope_sin_q": eng.rope[1][:N], "rope_cos_k": eng.rope[2][:N], "rope_sin_k": eng.rope[3][:N]}
    for bad in ([N + 1, 5], [0, 5]):
        pre["seq_len"] = torch.tensor(bad, dtype=torch.int32, device=DEV)
        with pytest.raises(RuntimeError, match="seq_len"):
            eng.transformer_steps(x, pre, 0, 1, seq_len_host=bad)
    a = rt.vv_gemm_args()
    A = torch.zeros(64, 64, device=DEV)
    a.dtype, a.out_dtype, a.mode = rt.VV_F32, rt.VV_F32, 0
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr() + 4, 64, A.data_ptr(), 64, A.data_ptr(), 64, 64, 128, 64
    assert eng.lib.vv_gemm(eng.ctx, C.byref(a), gu.stream()) == -22 and b"aligned" in eng.lib.vv_last_error(eng.ctx)
    pcm = torch.zeros(1, 64, dtype=torch.int16, device=DEV)
    assert eng.lib.vv_conv_post(eng.ctx, A.data_ptr(), A.data_ptr(), 0.0, pcm.data_ptr(), 64, None, 1, 8, 64, 5, 0.01, None, gu.stream()) == -22
    assert b"k=7" in eng.lib.vv_last_error(eng.ctx)
    # the context is still usable
    y = gu.gemm(eng, torch.ones(64, 64, device=DEV), torch.ones(128, 64, device=DEV))
    assert float(y.min()) == float(y.max()) == 64.0


def test_short_reference_clip_is_refused_with_host_lengths(hip_tiny, tiny_setup):
    """a reference clip of n_fft / 2 samples or fewer has no defined centred STFT (torch.stft refuses it; the reference
    admits any clip, /root/reference/vietvoicetts/core/audio_processor.py:15-26, core/tts_engine.py:46-56).  With the clip lengths on
    the host (vv_preprocess_h) the call returns -22 naming the item BEFORE anything is launched -- also when the short item sits beside
    a long one, the case that used to be silently double-reflected -- and the context keeps working; the device-only form stays finite."""
    spec, w, _ = tiny_setup
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(8)
    short = spec.n_fft // 2                                   # one sample too few
    la, lt, gen = [256 * 10, short], [12, 9], [6, 4]
    seq = [la[b] // spec.hop_length + 1 + gen[b] for b in range(2)]
    N = max(seq)
    audio = torch.zeros(2, max(la), dtype=torch.int16)
    for b in range(2):
        audio[b, : la[b]] = (torch.randn(la[b], generator=g) * 3000).to(torch.int16)
    ids = torch.randint(1, spec.vocab_size, (2, max(lt)), generator=g, dtype=torch.int32)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    args = (audio.to(DEV), i32(la), ids.to(DEV), i32(lt), i32(seq), N)
    with pytest.raises(RuntimeError, match=r"reference clip 1 has %d samples" % short):
        eng.preprocess(*args, audio_len_host=la)
    with pytest.raises(RuntimeError, match="reference clip 0 has"):       # a host length beyond max_audio_len is a caller bug too
        eng.preprocess(*args, audio_len_host=[max(la) + 1, 600])
    ok = eng.preprocess(*args)                                # device-only lengths: defined by the kernel's clamp (finite, deterministic)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ok["cat_mel_text"]).all())
    la2 = [la[0], short + 1]                                  # the smallest admitted clip
    ok2 = eng.preprocess(audio.to(DEV), i32(la2), ids.to(DEV), i32(lt), i32(seq), N, audio_len_host=la2)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ok2["cat_mel_text"]).all()) and int(ok2["ref_signal_len"][1]) == la2[1] // spec.hop_length + 1


def test_device_lengths_shorter_than_host_lengths_leave_no_stray_index(hip_tiny, tiny_setup):
    """the launch shapes come from the HOST copy of the lengths; a device copy that sums to FEWER rows used to leave
    row_src / row_pos of the missing rows unwritten (arena memory), and cfg_euler / pack_cat gather and scatter through them.  The last
    workgroup of row_tables_kernel now maps those rows onto padding rows of the last item.  A larger call runs first, so that the table
    offsets of the mismatched call hold activation bit patterns (huge as indices) rather than a previous call's valid tables.  Checked:
    the call returns, the state stays finite, and item 0 -- whose lengths agree -- equals the matched run bit for bit."""
    spec, w, _ = tiny_setup
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(21)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)

    def inputs(la, lt, gen):
        seq = [la[b] // spec.hop_length + 1 + gen[b] for b in range(len(la))]
        audio = torch.zeros(len(la), max(la), dtype=torch.int16)
        for b in range(len(la)):
            audio[b, : la[b]] = (torch.randn(la[b], generator=g) * 3000).to(torch.int16)
        ids = torch.randint(1, spec.vocab_size, (len(la), max(lt)), generator=g, dtype=torch.int32)
        return audio, ids, seq, torch.randn(len(la), max(seq), spec.n_mel, generator=g)

    la, lt = [256 * 8, 256 * 6], [10, 7]
    audio, ids, seq_host, noise = inputs(la, lt, [9, 7])
    seq_dev = [seq_host[0], seq_host[1] - 5]                 # the device array claims fewer rows for the LAST item
    N = max(seq_host)
    pre = eng.preprocess(audio.to(DEV), i32(la), ids.to(DEV), i32(lt), i32(seq_host), N, seq_len_host=seq_host)
    x_ok = noise.to(DEV).clone()
    eng.transformer_steps(x_ok, pre, 0, 2)
    # a larger call: the arena region where the next call's tables will sit now holds activations
    lb, ltb = [256 * 20, 256 * 18, 256 * 16], [20, 18, 16]
    ab, ib, sb, nb = inputs(lb, ltb, [30, 28, 26])
    pre_b = eng.preprocess(ab.to(DEV), i32(lb), ib.to(DEV), i32(ltb), i32(sb), max(sb), seq_len_host=sb)
    eng.transformer_steps(nb.to(DEV).clone(), pre_b, 0, 1)
    pre_bad = dict(pre, seq_len=i32(seq_dev))
    x_bad = noise.to(DEV).clone()
    eng.transformer_steps(x_bad, pre_bad, 0, 2, seq_len_host=seq_host)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x_bad).all())
    assert torch.equal(x_bad[0, : seq_host[0]], x_ok[0, : seq_host[0]])            # item 0 is untouched by item 1's mismatch


def test_conv_post_negative_length_is_an_empty_row(hip_tiny):
    """a negative len_in (reachable through the exported vv_conv_post) is an empty row, not a 4 GiB buffer resource."""
    from tests import gpu_util as gu
    eng = hip_tiny["f32"]
    B, Cc, T = 2, 8, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cc, T, generator=g)
    wgt = torch.randn(Cc, 7, generator=g) * 0.1
    pcm = torch.full((B, T), 77, dtype=torch.int16, device=DEV)
    wave = torch.full((B, T), 9.0, device=DEV)
    dl = torch.tensor([-5, T], dtype=torch.int32, device=DEV)
    bias = 0.25
    gu.check(eng, eng.lib.vv_conv_post(eng.ctx, x.to(DEV).data_ptr(), wgt.to(DEV).data_ptr(), bias, pcm.data_ptr(), T, wave.data_ptr(), B, Cc, T, 7, 0.01,
                                       dl.data_ptr(), gu.stream()))
    torch.cuda.synchronize()
    assert float((wave[0] - math.tanh(bias)).abs().max()) < 1e-6        # every input sample of the empty row reads as zero: tanh(bias)
    ref1 = torch.tanh(F.conv1d(F.leaky_relu(x[1:2], 0.01), wgt.reshape(1, Cc, 7), torch.tensor([bias]), padding=3)).reshape(T)
    assert float((wave[1].cpu() - ref1).abs().max()) < 2e-6

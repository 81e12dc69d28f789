"""-m gpu: BASELINE configs[3] -- "batch=256 mixed-length (64-512 tok) with reference-audio voice cloning, sharded across 8 x MI355X" --
as far as one GPU can show it: RANK 0's SHARD of the 256 seeded units (bench.make_mixed_inputs: 32 ragged units, LPT-sharded by frame
cost, per-item reference clips of 3-9 s, texts of 64-512 tokens) at FULL model size in configs[3]'s arithmetic (bf16 acoustic + fp32-
fidelity vocoder), all 31 Euler steps, through the call bench.py times.  The reference synthesises every (clip, chunk) unit as an
independent B = 1 call (/root/reference/vietvoicetts/core/tts_engine.py:111-122,225-238), so the property checked is exactly that:
every item of the packed ragged batch is finite and full length, and three items (shortest, longest, one in the middle) equal the
SAME item run alone -- BIT FOR BIT since round 4 (every kernel is row- or sequence-local and the two GEMM kernels share one arithmetic;
before that: the bf16 batch-vs-alone class, 5e-3).  The 8-GPU leg itself is unmeasured on hardware (no 8-GPU node).
PARITY UNPINNED against the real reference graphs (oracle/vv_oracle.py header)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DEV = "cuda:0"


def test_rank0_shard_of_configs3_bf16_every_item_and_three_alone():
    import bench
    from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
    from vietvoice_tts_amd.runtime import HipSynth
    spec = ModelSpec.full()
    w = make_synthetic_weights(spec, bench.SEED)
    batches, audio_s, n_units, fill = bench.make_mixed_inputs(spec, 32, 0, 8, DEV)
    assert n_units == 32 and len(batches) == 1, "rank 0 of 8 holds 32 units as one packed ragged batch"
    d, N, t_gen = batches[0]
    seq, gen = d["seq_len_host"], d["gen_frames"]
    la, lt = [int(v) for v in d["audio_len"].cpu()], [int(v) for v in d["text_len"].cpu()]
    assert min(lt) >= 64 and max(lt) <= 512 and min(la) >= 3 * 24000 - 256 and max(la) <= 9 * 24000 and max(seq) <= 1875 and len(set(seq)) > 16
    eng = HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=32)
    x, pcm, pcm_len, _pre = eng.synthesize_batch(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], N, d["noise"], t_gen,
                                                 gen_frames=gen, seq_len_host=seq, audio_len_host=la)
    torch.cuda.synchronize()
    hop = spec.hop_length
    for b in range(32):
        assert bool(torch.isfinite(x[b, : seq[b]]).all()), b
        assert int(pcm_len[b]) == gen[b] * hop, (b, int(pcm_len[b]), gen[b])
        assert int(pcm[b, : gen[b] * hop].abs().max()) > 0, b
    assert abs(float(pcm_len.sum()) / spec.sample_rate - audio_s) < 1e-3
    order = sorted(range(32), key=lambda b: seq[b])
    picks = [order[0], order[16], order[-1]]                   # shortest, one in the middle, longest
    for b in picks:
        one = eng.synthesize_batch(d["audio"][b:b + 1, : la[b]].contiguous(), d["audio_len"][b:b + 1].contiguous(),
                                   d["ids"][b:b + 1, : lt[b]].contiguous(), d["text_len"][b:b + 1].contiguous(), d["seq_len"][b:b + 1].contiguous(),
                                   seq[b], d["noise"][b:b + 1, : seq[b]].contiguous(), gen[b], seq_len_host=[seq[b]], audio_len_host=[la[b]])
        torch.cuda.synchronize()
        xa, pa = one[0][0], one[1][0, : gen[b] * hop]
        xb, pb = x[b, : seq[b]], pcm[b, : gen[b] * hop]
        rel = float((xa - xb).double().pow(2).mean().sqrt() / xb.double().pow(2).mean().sqrt())
        dp = (pa.int() - pb.int()).abs()
        print(f"\n[mixed256 rank-0 shard, full bf16, 31 steps] item {b} (N = {seq[b]}, T = {lt[b]}, clip {la[b] / 24000:.2f} s, gen {gen[b]} frames): "
              f"in the batch vs alone: state rmse/rms {rel:.3e}, PCM max diff {int(dp.max())} LSB")
        assert rel <= BATCH_VS_ALONE_STATE and int(dp.max()) <= BATCH_VS_ALONE_PCM_LSB, (b, rel, int(dp.max()))
    eng.close()


# measured when the test was written (round 4): state rmse/rms 0.000e+00 and 0 LSB on all three items -- the packed batch (persistent
# 256 x 256 GEMM) and the item alone (128 x 128 kernel) share one arithmetic, every other kernel is row- or sequence-local.  Held exact.
BATCH_VS_ALONE_STATE = 0.0
BATCH_VS_ALONE_PCM_LSB = 0

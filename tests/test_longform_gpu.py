"""-m gpu: BASELINE configs[4] at FULL model size under test -- "streaming long-form (4k-token) chunked synthesis, batch=8,
hipGraph-captured vocoder step": the per-chunk loop of /root/reference/vietvoicetts/core/tts_engine.py:225-246 (chunk plan,
per-chunk synthesis, improved cross-fade) through the drop-in TTSEngine with 8 chunks in flight and the decode stage replayed
from a captured hipGraph.  The text is bench.py's long-form text (4,096 characters).

Checked: the chunk count is the plan's; the buffered result (`synthesize`) with the captured vocoder equals the eager (no graph)
engine on the same seed; `synthesize_stream` with the same 8-chunk groups equals the buffered result; the second call on one
engine replays the cached graph (hit, no new capture).  bf16 acoustic: the chunks of one text are grouped by length in the
buffered call and in text order in the streamed one, i.e. a chunk shares its GEMM launches with different neighbours.  Rows are
packed and every kernel is row- or sequence-local, so streamed and buffered audio are equal to the cross-fade's LSB (measured 0):
a chunk's audio does not depend on what shares its launch.  The one exception is an OPTION, off by default since round 4: the
split-K tail of the FF2 GEMM (tail rows sum fp32 K parts in another order than the MFMA accumulator; the next bf16 rounding
turns that into bf16-level noise: measured 22 LSB max / 3.4e-4 rmse / rms on a +-20,000 LSB signal with the parts in fp32, 24 LSB
with the round-3 bf16 parts) -- checked here as the option's documented tolerance class.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SENT = "Hôm nay trời đẹp quá, chúng ta cùng nhau đi dạo quanh hồ và ngắm hoa nở bên đường nhé. "
TEXT = (SENT * 60)[:4096]


def _engine(tmp, split_k_tail=None, **kw):
    from vietvoice_tts_amd.core import ModelConfig, TTSEngine
    cfg = ModelConfig(model_cache_dir=str(tmp), synthetic_model=True, model_spec="full", acoustic_dtype="bf16", nfe_step=32,
                      max_batch_chunks=8, **kw)
    eng = TTSEngine(cfg)
    if split_k_tail is not None:
        eng.model_session_manager.engine.set_option("split_k_tail", split_k_tail)
    return eng


def _diff(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d > 1).mean()), float(np.sqrt((d.astype(np.float64) ** 2).mean()) / max(np.sqrt((a.astype(np.float64) ** 2).mean()), 1.0))


def test_longform_4k_chars_batch8_hipgraph_vocoder(tmp_path):
    assert len(TEXT) == 4096
    g = _engine(tmp_path, use_hip_graph=True)
    wg, secs = g.synthesize(TEXT)
    plan = list(g._last_plan)
    n_chunks = len(plan)
    cache = g._decode_graphs
    assert n_chunks >= 16 and wg.dtype == np.int16 and wg.ndim == 1 and secs > 0
    # the plan is the reference's arithmetic: every chunk fits the 20 s window
    assert all(f * 256 / 24000.0 <= g.config.max_chunk_duration + 0.02 for f in plan)
    n_groups = -(-n_chunks // 8)
    captured = cache.misses
    assert 1 <= captured <= n_groups and cache.hits + cache.misses == n_groups
    audio_s = wg.size / 24000.0
    print(f"\n[longform full bf16] {n_chunks} chunks in {n_groups} groups of <= 8, {audio_s:.1f} s of audio in {secs:.2f} s wall "
          f"({audio_s / secs:.1f} audio-s/s incl. capture), {captured} decode graph(s) captured, cache pins {cache.pinned_bytes() / 2**30:.2f} GiB")
    # second call on the same engine: every group replays a cached graph
    wg2, secs2 = g.synthesize(TEXT)
    assert cache.misses == captured and cache.hits >= 2 * n_groups - captured and wg2.shape == wg.shape
    print(f"[longform full bf16] second call {secs2:.2f} s wall ({audio_s / secs2:.1f} audio-s/s), graph cache hits {cache.hits} misses {cache.misses}")
    g.cleanup()

    # ---- streamed == buffered (same seed: fresh engine), 8 chunks per step like the buffered groups
    s = _engine(tmp_path, use_hip_graph=True)
    blocks = list(s.synthesize_stream(TEXT, chunks_per_step=8))
    ws = np.concatenate(blocks)
    assert len(blocks) == n_groups and list(s._last_plan) == plan
    s.cleanup()
    mx, share, rel = _diff(ws, wg)
    print(f"[longform full bf16] streamed vs buffered: max {mx} LSB, share beyond 1 LSB {share:.2e}, rmse/rms {rel:.2e}")

    # ---- captured vocoder == eager vocoder (same seed, same groups)
    e = _engine(tmp_path, use_hip_graph=False)
    we, _ = e.synthesize(TEXT)
    assert list(e._last_plan) == plan
    e.cleanup()
    mx_e, share_e, rel_e = _diff(we, wg)
    print(f"[longform full bf16] hipGraph vocoder vs eager: max {mx_e} LSB, share beyond 1 LSB {share_e:.2e}, rmse/rms {rel_e:.2e}")
    assert mx_e <= 1 and share_e <= 1e-4
    assert mx <= 2 and share <= 1e-4                    # default (no split-K tail): a row's arithmetic is independent of its batch neighbours

    # ---- the same comparison with the position-dependent split-K tail option ON (FF2): bf16 tolerance class, 3 x measured
    b2 = _engine(tmp_path, split_k_tail=2, use_hip_graph=True)
    w2, _ = b2.synthesize(TEXT)
    b2.cleanup()
    s2 = _engine(tmp_path, split_k_tail=2, use_hip_graph=True)
    ws2 = np.concatenate(list(s2.synthesize_stream(TEXT, chunks_per_step=8)))
    s2.cleanup()
    mx2, share2, rel2 = _diff(ws2, w2)
    print(f"[longform full bf16] split_k_tail = 2 (option): streamed vs buffered max {mx2} LSB, share beyond 1 LSB {share2:.2e}, rmse/rms {rel2:.2e}")
    assert mx2 <= 72 and rel2 <= 1e-3

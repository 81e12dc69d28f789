"""Does the headline batch gain from running as TWO half batches on two HIP streams at once (one's LayerNorm passes and kernel tails
beside the other's GEMMs)?  Diagnostic; no product code changes.

    python tools/overlap_probe.py [--batch 32] [--rounds 3] [--cu-split 0]

Legs, alternated `--rounds` times in one process on one box:
  whole      one context, B items, one stream                              (what bench.py times)
  serial     two contexts, B/2 items each, one after the other             (what halving the launches costs)
  overlap    two contexts, B/2 items each, two threads on two streams      (the question)
  masked     as overlap, on streams created with hipExtStreamCreateWithCUMask: context 0 on the first `--cu-split` CUs of every XCD
             pattern, context 1 on the rest (only with --cu-split > 0)
Every item's arithmetic is independent of its batch (tests/test_mixed256_gpu.py), so all legs produce the same PCM; checked.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HipSynth, sharding = bench.HipSynth, bench.sharding


def masked_stream(device, mask_words):
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(mask_words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--parts", type=int, nargs="+", default=[2])
    ap.add_argument("--prio", action="store_true", help="a leg with the two streams at different priorities")
    ap.add_argument("--cu-split", type=int, default=0, help="CUs (of 256) for context 0's masked stream; 0 = no masked leg")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = bench.ModelSpec.full()
    weights = bench.make_synthetic_weights(spec, bench.SEED)
    flat, _ = sharding.broadcast_weights(spec, torch.bfloat16, weights, dev)
    B = a.batch
    d, N = bench.make_inputs(spec, B, 0, dev)

    def half(lo, hi):
        h = {k: (v[lo:hi].contiguous() if torch.is_tensor(v) else v) for k, v in d.items() if k != "host"}
        h["seq_len_host"] = d["seq_len_host"][lo:hi]
        return h
    PMAX = max(a.parts)
    engs = [HipSynth(spec, None, device=str(dev), acoustic_dtype="bf16", nfe_step=32, flat_weights=flat) for _ in range(1 + PMAX)]

    def parts(P):
        cut = [B * i // P for i in range(P + 1)]
        return [half(cut[i], cut[i + 1]) for i in range(P)]
    halves = parts(2)

    def run(e, dd):
        return e.synthesize_batch(dd["audio"], dd["audio_len"], dd["ids"], dd["text_len"], dd["seq_len"], N, dd["noise"], bench.GEN_FRAMES,
                                  seq_len_host=dd["seq_len_host"])

    def whole():
        o = run(engs[0], d)
        torch.cuda.synchronize()
        return [o[1]]

    def serial():
        o = [run(engs[1], halves[0]), run(engs[2], halves[1])]
        torch.cuda.synchronize()
        return [x[1] for x in o]

    def overlapped(streams, pieces=None):
        pieces = halves if pieces is None else pieces
        out = [None] * len(pieces)

        def work(i):
            with torch.cuda.stream(streams[i]):
                out[i] = run(engs[1 + i], pieces[i])
                streams[i].synchronize()
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(pieces))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        return [x[1] for x in out]

    plain = [torch.cuda.Stream(dev) for _ in range(PMAX)]
    legs = {"whole": whole, "serial": serial}
    for P in a.parts:
        legs[f"overlap{P}"] = (lambda pc: (lambda: overlapped(plain, pc)))(parts(P))
    if a.prio:
        pr = [torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)]
        legs["prio2"] = lambda: overlapped(pr)
    if a.cu_split > 0:
        n0 = a.cu_split
        bits0 = sum(1 << i for i in range(256) if (i * n0) // 256 != ((i + 1) * n0) // 256)     # n0 CUs spread evenly over the index space
        bits1 = ((1 << 256) - 1) ^ bits0
        words = lambda b: [(b >> (32 * w)) & 0xFFFFFFFF for w in range(8)]
        ms = [masked_stream(dev, words(bits0)), masked_stream(dev, words(bits1))]
        legs["masked"] = lambda: overlapped(ms)
    audio_s = B * bench.GEN_FRAMES * spec.hop_length / spec.sample_rate
    ref = None
    for name, f in legs.items():                     # warm-up + equality
        pcm = torch.cat(f(), 0)
        if ref is None:
            ref = pcm
        print(f"[{name}] warm-up done; PCM equal to `whole`: {bool(torch.equal(pcm, ref))}", flush=True)
    res = {k: [] for k in legs}
    for r in range(a.rounds):
        for name, f in legs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f()
            dt = time.perf_counter() - t0
            res[name].append(dt)
            print(f"round {r} {name:8s} {dt * 1e3:8.1f} ms  {audio_s / dt:7.2f} audio-s/s", flush=True)
    print(json.dumps({k: {"ms": [round(x * 1e3, 1) for x in v], "best_audio_s_per_s": round(audio_s / min(v), 2)} for k, v in res.items()}))


if __name__ == "__main__":
    main()

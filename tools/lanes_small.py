"""Two lanes against one at small shapes (where does the host's launch rate eat the gain?): whole synthesize_batch calls, full model, bf16.
    python tools/lanes_small.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
spec = bench.ModelSpec.full()
flat, _ = bench.sharding.broadcast_weights(spec, torch.bfloat16, bench.make_synthetic_weights(spec, bench.SEED), dev)
eng = bench.HipSynth(spec, None, device=str(dev), acoustic_dtype="bf16", nfe_step=32, flat_weights=flat)
for B, ref_s, gen in [(2, 1.0, 60), (2, 2.0, 200), (2, 3.0, 400), (4, 1.0, 60), (4, 2.0, 200), (8, 1.0, 60), (8, 2.0, 200), (2, 6.0, 1037), (3, 6.0, 1037)]:
    bench.REF_SAMPLES, bench.GEN_FRAMES = int(ref_s * 24000), gen
    d, N = bench.make_inputs(spec, B, 0, dev)
    res = {}
    for lanes in (1, 2, 1, 2):
        eng.set_option("lanes", lanes)
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.synthesize_batch(d["audio"], d["audio_len"], d["ids"], d["text_len"], d["seq_len"], N, d["noise"], gen, seq_len_host=d["seq_len_host"])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res.setdefault(lanes, []).append(dt * 1e3)
    print(f"B={B} N={N} ({2 * B * N} packed rows): one lane {min(res[1]):7.1f} ms, two lanes {min(res[2]):7.1f} ms  ({(min(res[2]) / min(res[1]) - 1) * 100:+.1f} %)", flush=True)

#!/usr/bin/env python3
"""Vocoder (decode stage) microbenchmark through the C ABI at the benchmark shape: B=32, 1037 generated frames each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spec = ModelSpec.full()
# A/B of whole libraries in one process: VOC_AB_LIBS=libA.so,libB.so (interleaved rounds, fuse_mrf fixed to 0)
ab = [p for p in os.environ.get("VOC_AB_LIBS", "").split(",") if p]
if ab:
    w = make_synthetic_weights(spec)
    engs = []
    for p in ab:
        rt._lib = None
        rt._lib = rt.load_library(p)
        e = rt.HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=4)
        e.set_option("fuse_mrf", 0)
        engs.append(e)
    dev = "cuda:0"
    B, N, ref = 32, 1600, 563
    x = torch.randn(B, N, spec.n_mel, generator=torch.Generator().manual_seed(0)).to(dev)
    pre = {"ref_signal_len": torch.full((B,), ref, dtype=torch.int32, device=dev), "seq_len": torch.full((B,), N, dtype=torch.int32, device=dev)}
    outs = [e.decode(x, pre, N - ref)[0].clone() for e in engs]
    print("PCM identical across libraries:", all(torch.equal(o, outs[0]) for o in outs))
    for r in range(3):
        for p, e in zip(ab, engs):
            e.prof_enable(True)
            for _ in range(iters):
                e.decode(x, pre, N - ref)
            pr = e.prof_collect()["voc_conv"]
            e.prof_enable(False)
            print(f"{os.path.basename(p):28s} conv class {pr['ms'] / iters:.2f} ms  {pr['flops'] / (pr['ms'] * 1e-3) / 1e12:.1f} TFLOP/s f32", flush=True)
    sys.exit(0)
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"
B, N, ref = 32, 1600, 563
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, spec.n_mel, generator=g).to(dev)
pre = {"ref_signal_len": torch.full((B,), ref, dtype=torch.int32, device=dev), "seq_len": torch.full((B,), N, dtype=torch.int32, device=dev)}
for fuse in (1, 0, 1, 0):            # K12 fused MRF pairs (C <= 64 stages) vs two launches per pair, interleaved in one process
    eng.set_option("fuse_mrf", fuse)
    for _ in range(2):
        pcm, n = eng.decode(x, pre, N - ref)
    torch.cuda.synchronize()
    eng.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        pcm, n = eng.decode(x, pre, N - ref)
    e1.record(); torch.cuda.synchronize()
    p = eng.prof_collect()
    eng.prof_enable(False)
    print(f"fuse_mrf={fuse} decode B={B} frames={N - ref}: {e0.elapsed_time(e1) / iters:.2f} ms per batch; conv class {p['voc_conv']['ms'] / iters:.2f} ms in "
          f"{p['voc_conv']['launches'] // iters} launches, {p['voc_conv']['flops'] / (p['voc_conv']['ms'] * 1e-3) / 1e12:.1f} TFLOP/s f32; "
          f"checksum {int(pcm.int().abs().sum())}", flush=True)

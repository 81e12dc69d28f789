#!/usr/bin/env python3
"""Vocoder (decode stage) microbenchmark through the C ABI at the benchmark shape: B=32, 1037 generated frames each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spec = ModelSpec.full()
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

#!/usr/bin/env python3
"""K13 (conv_post + tanh + int16) alone at the headline decode shape (B = 32, C = 32, T = 265,472): time per launch and algorithmic GB/s
(4 C + 2 bytes per sample).  Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (tools/profile_conv_post_pmc.sh) it gives the kernel's
HBM-side traffic.   python tools/conv_post_bench.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spec = ModelSpec.tiny()
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"
B, Cc, T = 32, 32, 265472
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cc, T, generator=g).to(dev)
w = (torch.randn(Cc, 7, generator=g) * 0.05).to(dev)
pcm = torch.zeros(B, T, dtype=torch.int16, device=dev)
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)          # 256 MiB written between launches: the input never sits in the Infinity Cache
for _ in range(2):
    assert eng.lib.vv_conv_post(eng.ctx, x.data_ptr(), w.data_ptr(), 0.01, pcm.data_ptr(), T, None, B, Cc, T, 7, 0.01, None, st) == 0
torch.cuda.synchronize()
ts = []
for _ in range(n):
    flush.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.lib.vv_conv_post(eng.ctx, x.data_ptr(), w.data_ptr(), 0.01, pcm.data_ptr(), T, None, B, Cc, T, 7, 0.01, None, st)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
med = ts[len(ts) // 2]
by = B * T * (4.0 * Cc + 2.0)
print(f"conv_post B={B} C={Cc} T={T}: median {med * 1e3:.1f} us over {n} launches (min {ts[0] * 1e3:.1f}), algorithmic {by / 1e9:.3f} GB -> {by / med / 1e6:.0f} GB/s "
      f"= {by / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)

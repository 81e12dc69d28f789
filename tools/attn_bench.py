#!/usr/bin/env python3
"""Attention microbenchmark through the C ABI at the benchmark shape (64 sequences x 16 heads, N=1600, d=64)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
spec = ModelSpec.tiny()
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"
n_seq, N, H, D = 64, 1600, 16, 1024
g = torch.Generator().manual_seed(0)
qkv = torch.randn(n_seq * N, 3 * D, generator=g)
qkv[:, :D] *= 0.125
qkv = qkv.to(torch.bfloat16).to(dev)
out = torch.zeros(n_seq * N, D, dtype=torch.bfloat16, device=dev)
a = rt.vv_attn_args()
a.dtype, a.qkv, a.ld_qkv, a.out, a.ld_out, a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len = rt.VV_BF16, qkv.data_ptr(), 3 * D, out.data_ptr(), D, n_seq, N, H, D, None
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): assert eng.lib.vv_attention(eng.ctx, C.byref(a), st) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): eng.lib.vv_attention(eng.ctx, C.byref(a), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attention bf16 n_seq={n_seq} heads={H} N={N}: {ms*1e3:.1f} us  {4.0*n_seq*H*N*N*64/ms/1e9:.1f} TFLOP/s")

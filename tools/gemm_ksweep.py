#!/usr/bin/env python3
"""Fit per-tile time = a * nk + b for the persistent GEMM: sweep K at fixed M, N (plain bf16 store)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
spec = ModelSpec.tiny()
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator().manual_seed(0)
KS = [int(k) for k in sys.argv[3].split(',')] if len(sys.argv) > 3 else (256, 512, 1024, 2048, 4096)
for K in KS:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    a = rt.vv_gemm_args()
    a.dtype = a.out_dtype = rt.VV_BF16
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): eng.lib.vv_gemm(eng.ctx, C.byref(a), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.lib.vv_gemm(eng.ctx, C.byref(a), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tiles = ((M + 255) // 256) * (N // 256)
    rounds = -(-tiles // 256)
    print(f"K={K:5d} nk={K//64:3d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s   per-tile-round {ms*1e3/rounds:6.2f} us ({rounds} rounds)", flush=True)

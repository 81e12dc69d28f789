#!/usr/bin/env python3
"""Single-utterance GEMM shapes (M = 3,200 rows): the 128x128 kernel against the persistent 256x256 kernel, through the C ABI
(vv_gemm_args.tile).  Result (profiles/r02/gemm_b1_tiles.txt): the 128 tile wins 1.3-2.8x at this size, which is what the automatic
choice (256 from M >= 4096) already does."""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
spec = ModelSpec.tiny()
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"; M = 3200
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
cs = torch.rand(1600, 64, device=dev)
for name, mode, N, K, act in [("qkv", 1, 3072, 1024, 0), ("out", 3, 1024, 1024, 0), ("ff1", 0, 2048, 1024, 1), ("ff2", 3, 1024, 2048, 0)]:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    bias = torch.zeros(N, device=dev); gate = torch.ones(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    line = f"{name} M={M} N={N} K={K}:"
    for tile in (128, 256, 128, 256):
        a = rt.vv_gemm_args()
        a.dtype, a.out_dtype, a.mode, a.act = rt.VV_BF16, rt.VV_BF16, mode, act
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
        a.bias, a.gate = bias.data_ptr(), (gate.data_ptr() if mode == 3 else None)
        a.tile = tile
        if mode == 1:
            a.cos_q = a.sin_q = a.cos_k = a.sin_k = cs.data_ptr(); a.seq_n, a.rope_dim = 1600, 1024
            a.rope_cs_q = a.rope_cs_k = cs.data_ptr()
        for _ in range(3): assert eng.lib.vv_gemm(eng.ctx, C.byref(a), st) == 0, eng.lib.vv_last_error(eng.ctx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): eng.lib.vv_gemm(eng.ctx, C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        line += f"  tile{tile} {ms*1e3:6.1f} us ({2.0*M*N*K/ms/1e9:5.0f} TF/s)"
    print(line, flush=True)

#!/usr/bin/env python3
"""GEMM microbenchmark through the C ABI (the DiT block shapes at B=32): TFLOP/s per epilogue mode.

A/B builds of one kernel file: tools/build_variants.py + tools/gemm_ab.py (VVTTS_LIB selects a library here)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2] if len(sys.argv) > 2 else None
    spec = ModelSpec.tiny()
    eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
    dev = "cuda:0"
    M = 102400
    g = torch.Generator().manual_seed(0)
    shapes = [("qkv_rope", 1, 3072, 1024, 0), ("out_gate_store", 3, 1024, 1024, 0), ("ff1_gelu", 0, 2048, 1024, 1), ("ff2_gate_store", 3, 1024, 2048, 0),
              ("ff2_gate_res", 2, 1024, 2048, 0), ("plain_store", 0, 1024, 1024, 0)]
    cq = torch.rand(1600, 64, device=dev); 
    for name, mode, N, K, act in shapes:
        if only and only != name: continue
        A = (torch.randn(M, K, generator=g) * 1.0).to(torch.bfloat16).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, device=dev) * 0.1
        gate = torch.randn(N, device=dev)
        out = torch.zeros(M, N, dtype=torch.float32 if mode == 2 else torch.bfloat16, device=dev)
        a = rt.vv_gemm_args()
        a.dtype, a.out_dtype, a.mode, a.act = rt.VV_BF16, (rt.VV_F32 if mode == 2 else rt.VV_BF16), mode, act
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
        a.bias, a.gate = bias.data_ptr(), (gate.data_ptr() if mode >= 2 else None)
        if mode == 1:
            a.cos_q = a.sin_q = a.cos_k = a.sin_k = cq.data_ptr(); a.seq_n, a.rope_dim = 1600, 1024
            a.rope_cs_q = a.rope_cs_k = cq.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            assert eng.lib.vv_gemm(eng.ctx, C.byref(a), st) == 0, eng.lib.vv_last_error(eng.ctx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            eng.lib.vv_gemm(eng.ctx, C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:14s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()

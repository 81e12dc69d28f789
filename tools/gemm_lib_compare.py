#!/usr/bin/env python3
"""Yardstick from outside the repo: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches at the DiT block's four GEMM shapes
(M = 102,400, bf16, random operands, plain store without bias / activation / rope / gate), next to this repo's vv_gemm at the same shapes with
its fused epilogues.  Not a product dependency: a measurement of how close the persistent kernel is to a tuned library on this chip."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

dev = "cuda:0"
M = int(os.environ.get("GEMM_AB_M", 102400))
eng = rt.HipSynth(ModelSpec.tiny(), make_synthetic_weights(ModelSpec.tiny()), acoustic_dtype="bf16", nfe_step=4)
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for name, mode, N, K, act in (("qkv (rope in ours)", 1, 3072, 1024, 0), ("out (gate in ours)", 3, 1024, 1024, 0), ("ff1 (gelu in ours)", 0, 2048, 1024, 1), ("ff2 (gate in ours)", 3, 1024, 2048, 0)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    gate = torch.randn(N, generator=g).to(dev)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    a = rt.vv_gemm_args()
    a.dtype, a.out_dtype, a.mode, a.act = rt.VV_BF16, rt.VV_BF16, mode, act
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
    a.bias, a.gate = bias.data_ptr(), (gate.data_ptr() if mode == 3 else None)
    if mode == 1:
        cs = torch.rand(1600, 64, device=dev)
        a.cos_q = a.sin_q = a.cos_k = a.sin_k = cs.data_ptr(); a.seq_n, a.rope_dim, a.rope_theta = 1600, 1024, 10000.0
        a.rope_cs_q = a.rope_cs_k = cs.data_ptr()

    def ours():
        assert eng.lib.vv_gemm(eng.ctx, C.byref(a), st) == 0, eng.lib.vv_last_error(eng.ctx)
    Wt = W.t()

    def lib():
        torch.matmul(A, Wt, out=out)
    res = {}
    for rep in range(2):
        for nm, fn in (("ours", ours), ("library", lib)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(nm, []).append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    print(f"{name:20s} N={N} K={K}: ours {min(res['ours'])*1e3:7.1f} us ({fl/min(res['ours'])/1e9:6.0f} TF/s) | torch.matmul {min(res['library'])*1e3:7.1f} us ({fl/min(res['library'])/1e9:6.0f} TF/s)", flush=True)

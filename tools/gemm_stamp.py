#!/usr/bin/env python3
"""Where a K-tile phase of the persistent GEMM spends its cycles: runs DIAGNOSTIC builds (-DVV_GEMM_STAMP, tools/build_variants.py)
at the four DiT-block shapes (M = 102,400) and prints, per wave group, the average cycles per phase in the load segment (L), at the
barriers (B) and in the MFMA cluster (C), the epilogue per tile and the share of the kernel each takes.  The stamps' own waits
(s_memtime + lgkmcnt(0), ~40 cycles each, five per phase) change the timing: read the SHARES, never the length.
   python tools/gemm_stamp.py lib_stamp.so [lib_stamp_variant.so ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

libs = sys.argv[1:]
spec = ModelSpec.tiny()
w = make_synthetic_weights(spec)
dev = "cuda:0"
M = 102400
g = torch.Generator().manual_seed(0)
shapes = [("qkv_rope", 1, 3072, 1024, 0), ("out_gate_store", 3, 1024, 1024, 0), ("ff1_gelu", 0, 2048, 1024, 1), ("ff2_gate_store", 3, 1024, 2048, 0)]
cs = torch.rand(1600, 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for p in libs:
    rt._lib = None
    rt._lib = rt.load_library(p)
    e = rt.HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=4)
    print(f"== {os.path.basename(p)}")
    for name, mode, N, K, act in shapes:
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        if os.environ.get("GEMM_AB_ZERO"):      # power probe: same instruction stream on all-zero operands
            A.zero_(); W.zero_()
        bias = (torch.randn(N, generator=g) * 0.1).to(dev)
        gate = torch.randn(N, generator=g).to(dev)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        dbg = torch.zeros(256 * 2 * 10, dtype=torch.int64, device=dev)
        a = rt.vv_gemm_args()
        a.dtype, a.out_dtype, a.mode, a.act = rt.VV_BF16, rt.VV_BF16, mode, act
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
        a.bias, a.gate = bias.data_ptr(), (gate.data_ptr() if mode == 3 else None)
        a.C_tail = dbg.data_ptr()
        if mode == 1:
            a.cos_q = a.sin_q = a.cos_k = a.sin_k = cs.data_ptr(); a.seq_n, a.rope_dim = 1600, 1024
            a.rope_cs_q = a.rope_cs_k = cs.data_ptr()
            a.rope_skip_q = 1 if os.environ.get("GEMM_STAMP_SKIP_Q") else 0      # round 4: the q columns are roped by the attention kernel
            a.rope_theta = float(os.environ.get("GEMM_STAMP_THETA", "0"))         # round 4: > 0 = cos / sin computed in the epilogue (no tables)
        for _ in range(20):
            assert e.lib.vv_gemm(e.ctx, C.byref(a), st) == 0, e.lib.vv_last_error(e.ctx)
        torch.cuda.synchronize()
        d = dbg.view(256, 2, 10).cpu().double()
        for grp in (0, 1):
            v = d[:, grp]
            n, tiles, tot = v[:, 4].mean(), v[:, 6].mean(), v[:, 5].mean()
            L, B, Cc, E = v[:, 0].sum() / v[:, 4].sum(), v[:, 1].sum() / v[:, 4].sum(), v[:, 2].sum() / v[:, 4].sum(), v[:, 3].sum() / v[:, 6].sum()
            SW = v[:, 7].long()
            S, Wt = (SW & 0xffffffff).double().sum() / v[:, 6].sum(), (SW >> 32).double().sum() / v[:, 6].sum()
            clk = float((v[:, 5] / v[:, 8].clamp(min=1)).median()) * 0.1
            print(f"{name:15s} group {grp}: clock {clk:.2f} GHz (s_memtime / s_memrealtime) | per tile set-up {S:6.0f}  wait before the epilogue {Wt:6.0f} | per phase L {L:6.0f}  barriers {B:6.0f}  cluster {Cc:6.0f}  (sum {L + B + Cc:6.0f}) | epilogue per tile {E:7.0f} | "
                  f"kernel {tot:9.0f} cycles, {tiles:.2f} tiles/block, phases {n:.0f}: loop {100 * (v[:, 0] + v[:, 1] + v[:, 2]).mean() / tot:4.1f} %  epilogue {100 * v[:, 3].mean() / tot:4.1f} %", flush=True)
    e.close()

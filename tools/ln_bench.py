#!/usr/bin/env python3
"""LayerNorm+residual microbenchmark through the C ABI at the benchmark shape (R = 102400 rows, D = 1024, bf16 deltas / output)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
spec = ModelSpec.tiny()
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"
R, D = 102400, 1024
x = torch.randn(R, D, device=dev)
d1 = torch.randn(R, D, device=dev).bfloat16()
d2 = torch.randn(R, D, device=dev).bfloat16()
y = torch.zeros(R, D, device=dev, dtype=torch.bfloat16)
w = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(two, keep):
    a = rt.vv_ln_args()
    a.out_dtype = rt.VV_BF16
    a.x, a.ldx, a.y, a.ldy, a.R, a.D, a.w, a.b, a.add_one, a.eps = x.data_ptr(), D, y.data_ptr(), D, R, D, w.data_ptr(), b.data_ptr(), 1, 1e-6
    a.delta, a.delta_dtype, a.ld_delta = d1.data_ptr(), rt.VV_BF16, D
    a.delta2 = d2.data_ptr() if two else None
    a.keep_x = keep
    for _ in range(3): assert eng.lib.vv_layernorm(eng.ctx, C.byref(a), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): eng.lib.vv_layernorm(eng.ctx, C.byref(a), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    byt = R * D * ((4 + 2 + (2 if two else 0)) + (0 if keep else 4) + 2)
    print(f"ln two_deltas={two} keep_x={keep}: {ms*1e3:.1f} us  {byt/ms/1e6:.0f} GB/s")
run(True, 0); run(False, 1)

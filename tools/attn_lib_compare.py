#!/usr/bin/env python3
"""Outside yardstick for K8: torch's scaled_dot_product_attention (whatever fused backend this ROCm build ships: AOTriton / CK flash) at the
headline shape -- 64 sequences (32 items x 2 CFG branches) x 16 heads x 1,600 keys x head_dim 64, bf16, random data -- next to vv_attention
on the packed qkv buffer of the same shape.  A measurement only: the product never calls it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

dev = "cuda:0"
B, H, N, D = int(os.environ.get("ATTN_B", 64)), 16, 1600, 1024
eng = rt.HipSynth(ModelSpec.tiny(), make_synthetic_weights(ModelSpec.tiny()), acoustic_dtype="bf16", nfe_step=4)
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
out = torch.zeros(B * N, D, dtype=torch.bfloat16, device=dev)
a = rt.vv_attn_args()
a.dtype, a.qkv, a.ld_qkv, a.out, a.ld_out, a.n_seq, a.seq_n, a.heads, a.dim = rt.VV_BF16, qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, N, H, D
a.q_scale = 0.125
st = torch.cuda.current_stream().cuda_stream
q = qkv[:, :D].reshape(B, N, H, 64).permute(0, 2, 1, 3).contiguous()
k = qkv[:, D:2 * D].reshape(B, N, H, 64).permute(0, 2, 1, 3).contiguous()
v = qkv[:, 2 * D:].reshape(B, N, H, 64).permute(0, 2, 1, 3).contiguous()


def ours():
    assert eng.lib.vv_attention(eng.ctx, C.byref(a), st) == 0, eng.lib.vv_last_error(eng.ctx)


def lib():
    return F.scaled_dot_product_attention(q, k, v)


ref = lib().permute(0, 2, 1, 3).reshape(B * N, D).float()
ours(); torch.cuda.synchronize()
print("max |ours - library| / max |library| =", float((out.float() - ref).abs().max() / ref.abs().max()))
fl = 4.0 * B * H * N * N * 64
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
print(f"attention {B} x {H} heads x {N} x 64: vv_attention {min(res['ours'])*1e3:7.1f} us ({fl/min(res['ours'])/1e9:6.0f} TF/s) | torch SDPA {min(res['library'])*1e3:7.1f} us ({fl/min(res['library'])/1e9:6.0f} TF/s)")

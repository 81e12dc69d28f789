#!/usr/bin/env python3
"""Attention A/B at the benchmark shape: every library given on the command line is loaded into ONE process and timed in
interleaved rounds (cdna guide rule 24); outputs are compared with the first library's.
    python tools/attn_ab.py [rounds] lib1.so lib2.so ..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

rounds = int(sys.argv[1])
libs = sys.argv[2:]
spec = ModelSpec.tiny()
w = make_synthetic_weights(spec)
dev = "cuda:0"
n_seq, N, H, D = 64, 1600, 16, 1024
g = torch.Generator().manual_seed(0)
qkv = torch.randn(n_seq * N, 3 * D, generator=g)
qkv[:, :D] *= 0.125
qkv = qkv.to(torch.bfloat16).to(dev)
if os.environ.get("ATTN_AB_ZERO"):          # power probe: the same instruction stream on all-zero q / k / v (uniform softmax, no redo path after tile 0)
    qkv.zero_()
engs, outs = [], []
for p in libs:
    rt._lib = None
    lib = rt.load_library(p)
    rt._lib = lib
    e = rt.HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=4)
    engs.append(e)
    outs.append(torch.zeros(n_seq * N, D, dtype=torch.bfloat16, device=dev))
st = torch.cuda.current_stream().cuda_stream
args = []
for e, o in zip(engs, outs):
    a = rt.vv_attn_args()
    a.dtype, a.qkv, a.ld_qkv, a.out, a.ld_out, a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len = rt.VV_BF16, qkv.data_ptr(), 3 * D, o.data_ptr(), D, n_seq, N, H, D, None
    args.append(a)
    for _ in range(2):
        assert e.lib.vv_attention(e.ctx, C.byref(a), st) == 0, e.lib.vv_last_error(e.ctx)
torch.cuda.synchronize()
for i, o in enumerate(outs[1:], 1):
    d = (o.float() - outs[0].float()).abs().max().item()
    print(f"{os.path.basename(libs[i])}: max |out - out[0]| = {d:.3e}", flush=True)
times = [[] for _ in libs]
for r in range(rounds):
    for i, (e, a) in enumerate(zip(engs, args)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            e.lib.vv_attention(e.ctx, C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 5)
fl = 4.0 * n_seq * H * N * N * 64
for p, t in zip(libs, times):
    t = sorted(t)
    print(f"{os.path.basename(p):28s} median {t[len(t)//2]*1e3:7.1f} us  min {t[0]*1e3:7.1f} us  {fl/t[len(t)//2]/1e9:7.1f} TFLOP/s", flush=True)

#!/usr/bin/env python3
"""GEMM A/B at the DiT block shapes (M = 102,400): every library on the command line is loaded into ONE process, timed in
interleaved rounds and checked against the first library's output.   python tools/gemm_ab.py [rounds] lib1.so lib2.so ...
POSITION BIAS: the library timed first in a round runs 1-2.5 % slower than the ones after it (identical code measured 184.9 vs 180.2 us,
profiles/r02/gemm_ab_ngroup_first.txt).  For differences of that size give every library in both positions (A B A B) and compare
like with like, or confirm with alternated bench.py runs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

rounds = int(sys.argv[1])
libs = sys.argv[2:]
# a library may be given as path@KEY=VAL,KEY=VAL: the environment settings are applied before each of ITS calls (diagnostic builds that
# read their switches from the environment, e.g. libvvtts_exp.so@VV_GEMM_NGROUP=4,VV_GEMM_A_NT=1) -- one library, several settings
envs = []
for i, p in enumerate(libs):
    path, _, kv = p.partition("@")
    envs.append(dict(x.split("=", 1) for x in kv.split(",") if x))
    libs[i] = path
labels = [os.path.basename(p)[9:-3] + ("@" + ",".join(f"{k[8:]}={v}" for k, v in e.items()) if e else "") for p, e in zip(libs, envs)]
ENV_KEYS = sorted({k for e in envs for k in e})


def use_env(i):
    for k in ENV_KEYS:
        if k in envs[i]:
            os.environ[k] = envs[i][k]
        else:
            os.environ.pop(k, None)
spec = ModelSpec.tiny()
w = make_synthetic_weights(spec)
dev = "cuda:0"
engs = []
for p in libs:
    rt._lib = None
    rt._lib = rt.load_library(p)
    engs.append(rt.HipSynth(spec, w, acoustic_dtype="bf16", nfe_step=4))
M = int(os.environ.get("GEMM_AB_M", 102400))          # 3200 = one utterance (1,600 frames x 2 CFG branches)
g = torch.Generator().manual_seed(0)
shapes = [("qkv_rope", 1, 3072, 1024, 0), ("qkv_rope_rows", 1, 3072, 1024, 0), ("out_gate_store", 3, 1024, 1024, 0), ("ff1_gelu", 0, 2048, 1024, 1), ("ff2_gate_store", 3, 1024, 2048, 0),
          # epilogue-cost probes (only with GEMM_AB_SHAPES): the QKV / FF1 shapes with the plain store epilogue
          ("qkv_plain_store", 0, 3072, 1024, 0), ("ff1_plain_store", 0, 2048, 1024, 0),
          ("qkv_rope_off_probe", 1, 3072, 1024, 0)]      # the rope-mode kernel with rope_dim = 0: every tile takes its plain store path
cs = torch.rand(1600, 64, device=dev)
pos = (torch.arange(M, dtype=torch.int32) % 1600).to(dev)
cs_rows = cs[pos.long()].contiguous()            # what vv_rope_rows builds once per call
st = torch.cuda.current_stream().cuda_stream
tot = [0.0 for _ in libs]
tiles = [int(x) for x in os.environ.get("GEMM_AB_TILES", "").split(",") if x]      # per library: vv_gemm_args.tile (0 auto, 128, 256), e.g. one library twice as 0,128
only = [x for x in os.environ.get("GEMM_AB_SHAPES", "").split(",") if x]      # e.g. the four shapes of one DiT block for a PMC pass
for name, mode, N, K, act in shapes:
    if (only and name not in only) or (not only and (name.endswith("_plain_store") or name.endswith("_probe"))):
        continue
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    if os.environ.get("GEMM_AB_ZERO"):          # power probe: all-zero operands toggle (almost) nothing in the matrix pipe -- same instruction stream, lower power
        A.zero_(); W.zero_()
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    gate = torch.randn(N, generator=g).to(dev)
    outs, args = [], []
    for li, e in enumerate(engs):
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        a = rt.vv_gemm_args()
        a.dtype, a.out_dtype, a.mode, a.act = rt.VV_BF16, rt.VV_BF16, mode, act
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
        a.bias, a.gate = bias.data_ptr(), (gate.data_ptr() if mode == 3 else None)
        if tiles:
            a.tile = tiles[li]
        if mode == 1:
            a.cos_q = a.sin_q = a.cos_k = a.sin_k = cs.data_ptr(); a.seq_n, a.rope_dim = 1600, (0 if name.endswith("_off_probe") else 1024)
            a.rope_cs_q = a.rope_cs_k = cs.data_ptr(); a.rope_pos = pos.data_ptr()
            if name.endswith("_rows") and hasattr(a, "rope_by_row"):
                a.rope_cs_q = a.rope_cs_k = cs_rows.data_ptr(); a.rope_by_row = 1
        use_env(li)
        for _ in range(2):
            assert e.lib.vv_gemm(e.ctx, C.byref(a), st) == 0, e.lib.vv_last_error(e.ctx)
        outs.append(out); args.append(a)
    torch.cuda.synchronize()
    diffs = [float((o.float() - outs[0].float()).abs().max()) for o in outs[1:]]
    if os.environ.get("GEMM_AB_REF"):          # ground truth on the first and the last 4096 rows (torch fp32 on the device)
        errs = []
        for o in outs:
            e_max = 0.0
            for lo in sorted({0, max(M - 4096, 0)}):
                R_ = min(4096, M - lo)
                ref = A[lo:lo + R_].float() @ W.float().t() + bias
                if act == 1:
                    ref = torch.nn.functional.gelu(ref, approximate="tanh")
                if mode == 3:
                    ref = ref * gate
                if mode == 1 and not name.endswith("_off_probe"):
                    p_ = pos[lo:lo + R_].long()
                    t = cs[p_]                                   # [rows][64] = (cos, sin) pairs of the compact table
                    for part in (0, 1):                          # q and k columns
                        x = ref[:, part * 1024:(part + 1) * 1024].reshape(R_, 16, 32, 2)
                        c, sn = t[:, 0::2].reshape(R_, 1, 32), t[:, 1::2].reshape(R_, 1, 32)
                        y = torch.stack((x[..., 0] * c - x[..., 1] * sn, x[..., 1] * c + x[..., 0] * sn), dim=-1)
                        ref[:, part * 1024:(part + 1) * 1024] = y.reshape(R_, 1024)
                e_max = max(e_max, float((o[lo:lo + R_].float() - ref).abs().max()))
            errs.append(round(e_max, 4))
        print(f"{name:15s} max |err| vs torch fp32 on 8192 rows: {errs}", flush=True)
    times = [[] for _ in libs]
    for r in range(rounds):
        for i, (e, a) in enumerate(zip(engs, args)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            use_env(i)
            e0.record()
            for _ in range(10):
                e.lib.vv_gemm(e.ctx, C.byref(a), st)
            e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10)
    line = f"{name:15s} N={N} K={K}:"
    for i, p in enumerate(libs):
        t = sorted(times[i]); med = t[len(t) // 2]
        if name != "qkv_rope":
            tot[i] += med
        line += f"  {labels[i]}{('/t%d' % tiles[i]) if tiles else ''} {med*1e3:6.1f} us ({2.0*M*N*K/med/1e9:6.0f} TF/s)"
    print(line + f"  | max diff vs first {diffs}", flush=True)
print("sum of qkv_rope_rows + out + ff1 + ff2 (one DiT block): " + "  ".join(f"{labels[i]} {tot[i]*1e3:.1f} us" for i in range(len(libs))), flush=True)

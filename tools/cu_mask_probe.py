"""How do the block's kernels run on a SUBSET of the CUs (hipExtStreamCreateWithCUMask), alone and beside a LayerNorm on the
complementary CUs?  Needs the diagnostic GEMM build (VV_GEMM_GRID):  python tools/build_variants.py vv_gemm exp=-DVV_GEMM_EXP
    python tools/cu_mask_probe.py vietvoice-tts_amd/libvvtts_exp.so 256 224 200 192
For every CU count C: GEMM shapes (M = 102,400 and 51,200 rows) and attention on a stream masked to the first C CU bits with grid = C
(alone), LayerNorm on the other 256 - C (alone), then both at once from two threads."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

lib_path = sys.argv[1]
counts = [int(x) for x in sys.argv[2:]] or [256, 224, 200]
rt._lib = rt.load_library(lib_path)
spec = ModelSpec.tiny()
eng = rt.HipSynth(spec, make_synthetic_weights(spec), acoustic_dtype="bf16", nfe_step=4)
dev = "cuda:0"
hip = C.CDLL("libamdhip64.so")


def masked(bits):
    s = C.c_void_p()
    words = [(bits >> (32 * w)) & 0xFFFFFFFF for w in range(8)]
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, (C.c_uint32 * 8)(*words))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


g = torch.Generator().manual_seed(0)
MMAX = 102400
A1 = torch.randn(MMAX, 1024, generator=g).to(torch.bfloat16).to(dev)
A2 = torch.randn(MMAX, 2048, generator=g).to(torch.bfloat16).to(dev)
Ws = {(n, k): (torch.randn(n, k, generator=g) * 0.03).to(torch.bfloat16).to(dev) for n, k in ((3072, 1024), (1024, 1024), (2048, 1024), (1024, 2048))}
bias = (torch.randn(3072, generator=g) * 0.1).to(dev)
gate = torch.randn(3072, generator=g).to(dev)
Cout = torch.empty(MMAX, 3072, dtype=torch.bfloat16, device=dev)
xres = torch.randn(MMAX, 1024, generator=g).to(dev)
d1 = torch.randn(MMAX, 1024, generator=g).to(torch.bfloat16).to(dev)
d2 = torch.randn(MMAX, 1024, generator=g).to(torch.bfloat16).to(dev)
hout = torch.empty(MMAX, 1024, dtype=torch.bfloat16, device=dev)
mod = torch.randn(2048, generator=g).to(dev)
dummy = torch.rand(1600, 64, device=dev)


def gemm_call(M, N, K, mode, act):
    a = rt.vv_gemm_args()
    a.dtype = a.out_dtype = rt.VV_BF16
    a.mode, a.act = mode, act
    A = A1 if K == 1024 else A2
    W = Ws[(N, K)]
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = A.data_ptr(), K, W.data_ptr(), K, Cout.data_ptr(), N
    a.M, a.N, a.K = M, N, K
    a.bias = bias.data_ptr()
    if mode == 3:
        a.gate = gate.data_ptr()
    if mode == 1:
        a.cos_q = a.sin_q = a.cos_k = a.sin_k = dummy.data_ptr()
        a.seq_n, a.rope_dim, a.rope_theta = 1600, 1024, 10000.0
    return lambda st: eng.lib.vv_gemm(eng.ctx, C.byref(a), st)


def ln_call(M):
    a = rt.vv_ln_args()
    a.out_dtype = rt.VV_BF16
    a.x, a.ldx, a.y, a.ldy, a.R, a.D = xres.data_ptr(), 1024, hout.data_ptr(), 1024, M, 1024
    a.w, a.b, a.add_one, a.eps = mod.data_ptr(), mod.data_ptr() + 4096, 1, 1e-6
    a.delta, a.delta_dtype, a.ld_delta, a.delta2, a.keep_x = d1.data_ptr(), rt.VV_BF16, 1024, d2.data_ptr(), 0
    return lambda st: eng.lib.vv_layernorm(eng.ctx, C.byref(a), st)


def attn_call(nseq):
    kv = torch.full((nseq,), 1600, dtype=torch.int32, device=dev)
    rs = (torch.arange(nseq, dtype=torch.int32) * 1600).to(dev)
    a = rt.vv_attn_args()
    a.dtype, a.qkv, a.ld_qkv, a.out, a.ld_out = rt.VV_BF16, Cout.data_ptr(), 3072, hout.data_ptr(), 1024
    a.n_seq, a.seq_n, a.heads, a.dim, a.total_rows, a.q_scale = nseq, 1600, 16, 1024, nseq * 1600, 0.125
    a.kv_len, a.row_start = kv.data_ptr(), rs.data_ptr()
    a._keep = (kv, rs)
    return lambda st: eng.lib.vv_attention(eng.ctx, C.byref(a), st)


def timed(fn, stream, n):
    """n launches back to back on `stream`; us per launch"""
    for _ in range(3):
        assert fn(stream.cuda_stream) == 0, eng.lib.vv_last_error(eng.ctx)
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(stream.cuda_stream)
    stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


Cout.normal_()
jobs = [("qkv 102400", gemm_call(102400, 3072, 1024, 1, 0), 645e9), ("out 102400", gemm_call(102400, 1024, 1024, 3, 0), 215e9),
        ("ff1 102400", gemm_call(102400, 2048, 1024, 0, 1), 430e9), ("ff2 102400", gemm_call(102400, 1024, 2048, 3, 0), 430e9),
        ("qkv  51200", gemm_call(51200, 3072, 1024, 1, 0), 322e9), ("out  51200", gemm_call(51200, 1024, 1024, 3, 0), 107e9),
        ("ff1  51200", gemm_call(51200, 2048, 1024, 0, 1), 215e9), ("ff2  51200", gemm_call(51200, 1024, 2048, 3, 0), 215e9),
        ("attn 64 seq", attn_call(64), 4.0 * 64 * 16 * 1600 * 1600 * 64), ("attn 32 seq", attn_call(32), 4.0 * 32 * 16 * 1600 * 1600 * 64)]
ln_full, ln_half = ln_call(102400), ln_call(51200)
LN_BYTES = 14.0 * 1024          # per row: x r+w fp32, two bf16 deltas, bf16 out

for Cn in counts:
    os.environ["VV_GEMM_GRID"] = str(Cn)
    cs = masked((1 << Cn) - 1) if Cn < 256 else torch.cuda.Stream(dev)
    ms = masked(((1 << 256) - 1) ^ ((1 << Cn) - 1)) if Cn < 256 else None
    print(f"=== {Cn} CUs for the matrix kernels, {256 - Cn} for the LayerNorm", flush=True)
    if ms is not None:
        t = timed(ln_half, ms, 20)
        print(f"  LayerNorm (51,200 rows, both deltas, rewrite) alone on {256 - Cn} CUs: {t:8.1f} us  {51200 * LN_BYTES / t / 1e6:7.2f} TB/s", flush=True)
    else:
        t = timed(ln_half, cs, 20)
        print(f"  LayerNorm (51,200 rows) alone on all CUs: {t:8.1f} us  {51200 * LN_BYTES / t / 1e6:7.2f} TB/s", flush=True)
    for name, fn, fl in jobs:
        t_alone = timed(fn, cs, 12)
        line = f"  {name}: alone {t_alone:8.1f} us {fl / t_alone / 1e6:7.1f} TF/s"
        if ms is not None:
            # both at once: the LayerNorm loops on its CUs until the matrix loop is done
            stop = [False]
            cnt = [0]

            def spin():
                while not stop[0]:
                    for _ in range(4):
                        ln_half(ms.cuda_stream)
                    ms.synchronize()
                    cnt[0] += 4
            th = threading.Thread(target=spin)
            t0 = time.perf_counter()
            th.start()
            time.sleep(0.002)
            t_with = timed(fn, cs, 12)
            stop[0] = True
            th.join()
            wall = time.perf_counter() - t0
            line += f" | beside the LayerNorm {t_with:8.1f} us {fl / t_with / 1e6:7.1f} TF/s; LayerNorm meanwhile {wall / max(cnt[0], 1) * 1e6:7.1f} us per launch"
        print(line, flush=True)

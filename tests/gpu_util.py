"""Helpers for the -m gpu parity tests: call single kernels through the C ABI."""
import ctypes as C

import torch

from vietvoice_tts_amd import runtime as rt

DEV = "cuda:0"


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(eng, rc):
    assert rc == 0, eng.lib.vv_last_error(eng.ctx).decode()


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def gemm_tail_plan(eng, M, N, K):
    row0, parts = C.c_int32(0), C.c_int32(0)
    check(eng, eng.lib.vv_gemm_tail_plan(eng.ctx, M, N, K, C.byref(row0), C.byref(parts)))
    return row0.value, parts.value


def gemm(eng, A, W, bias=None, mode=0, act=0, out_dtype=None, gate=None, C_io=None, ropes=None, seq_n=0, rope_dim=0, n_store=0, tile=0,
         rope_pos=None, rope_by_row=0, tail=None, c_fill=0.0, rope_skip_q=0, rope_theta=0.0, chip_share=0):
    """A [M,K], W [N,K] on device, same dtype (bf16 or f32).  tail = (fp32 C_tail tensor [parts][M - row0][N], row0, parts): the
    split-K tail request.  c_fill: what a fresh output buffer holds before the launch (shows rows the kernel leaves unwritten)."""
    dt = rt.VV_BF16 if A.dtype == torch.bfloat16 else rt.VV_F32
    od = dt if out_dtype is None else out_dtype
    M, K = A.shape
    N = W.shape[0]
    if C_io is None:
        C_io = torch.full((M, N), c_fill, dtype=torch.bfloat16 if od == rt.VV_BF16 else torch.float32, device=DEV)
    a = rt.vv_gemm_args()
    a.dtype, a.out_dtype, a.mode, a.act = dt, od, mode, act
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), C_io.data_ptr(), C_io.stride(0)
    a.M, a.N, a.K = M, N, K
    a.bias = None if bias is None else bias.data_ptr()
    a.gate = None if gate is None else gate.data_ptr()
    if ropes is not None:
        a.cos_q, a.sin_q, a.cos_k, a.sin_k = [t.data_ptr() for t in ropes[:4]]
        if len(ropes) == 6:
            a.rope_cs_q, a.rope_cs_k = ropes[4].data_ptr(), ropes[5].data_ptr()
    a.n_store, a.seq_n, a.rope_dim, a.tile = n_store, seq_n, rope_dim, tile
    a.rope_pos = None if rope_pos is None else rope_pos.data_ptr()
    a.rope_by_row = rope_by_row
    a.rope_skip_q = rope_skip_q
    a.chip_share = chip_share
    a.rope_theta = rope_theta
    if tail is not None:
        a.C_tail, a.tail_row0, a.tail_parts = tail[0].data_ptr(), tail[1], tail[2]
    check(eng, eng.lib.vv_gemm(eng.ctx, C.byref(a), stream()))
    torch.cuda.synchronize()
    return C_io

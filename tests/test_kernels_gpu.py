"""-m gpu: every HIP kernel, called through the C ABI, against plain-torch fp32 references of the
same op on the same seeded inputs.  Tolerances: fp32 kernels 2e-4 of the output range (different
summation order only); bf16 kernels are compared with an fp32 reference fed the SAME bf16-rounded
operands, 1.5e-2 of the output range (bf16 output rounding + fp32 accumulation order)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_F32 = 2e-4
TOL_BF16 = 1.5e-2


def _imports():
    from vietvoice_tts_amd import runtime as rt
    from tests import gpu_util as gu
    return rt, gu


def _tol(dt):
    return TOL_BF16 if dt == torch.bfloat16 else TOL_F32


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (128, 128, 64), (1000, 384, 192), (31, 128, 256)])
def test_gemm_store_bias_act(hip_tiny, dtype, M, N, K):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g) * 0.1
    for act, fn in ((0, lambda x: x), (1, lambda x: F.gelu(x, approximate="tanh")), (2, F.gelu), (3, F.silu)):
        ref = fn(A.float() @ W.float().t() + b)
        got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), act=act)
        assert gu.rel_err(got, ref) < _tol(dtype), (act, gu.rel_err(got, ref))
    if dtype == torch.bfloat16:   # bf16 operands, fp32 output
        got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), out_dtype=rt.VV_F32)
        assert gu.rel_err(got, A.float() @ W.float().t() + b) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nstore_and_asymmetric(hip_tiny, dtype):
    """A = I-like with an ASYMMETRIC W catches a transposed C write; n_store masks padded columns."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    M, N, K = 128, 128, 128
    A = torch.eye(M, K).to(dtype)
    W = (torch.arange(N).float()[:, None] * 0.01 + torch.arange(K).float()[None, :] * 0.5).to(dtype)
    C0 = torch.full((M, N), 7.0, dtype=torch.float32, device=gu.DEV)
    got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), out_dtype=rt.VV_F32, C_io=C0, n_store=100)
    ref = A.float() @ W.float().t()
    assert gu.rel_err(got[:, :100], ref[:, :100]) < 1e-6
    assert torch.all(got[:, 100:].cpu() == 7.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_gate_residual(hip_tiny, dtype):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    M, N, K = 260, 256, 128
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b, gate = torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g)
    x0 = torch.randn(M, N, generator=g)
    ref = x0 + gate * (A.float() @ W.float().t() + b)
    got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=2, out_dtype=rt.VV_F32, gate=gate.to(gu.DEV),
                  C_io=x0.clone().to(gu.DEV))
    assert gu.rel_err(got, ref) < (2e-3 if dtype == torch.bfloat16 else TOL_F32)
    # gated store: the delta that the next LayerNorm adds to the residual stream
    dl = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=3, gate=gate.to(gu.DEV))
    assert gu.rel_err(dl, gate * (A.float() @ W.float().t() + b)) < _tol(dtype)
    ref1 = x0 + (A.float() @ W.float().t() + b)
    got1 = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=2, out_dtype=rt.VV_F32, C_io=x0.clone().to(gu.DEV))
    assert gu.rel_err(got1, ref1) < (2e-3 if dtype == torch.bfloat16 else TOL_F32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_qkv_rope(hip_tiny, dtype, tiny_setup):
    rt, gu = _imports()
    from oracle.vv_oracle import Oracle
    spec, _, orc = tiny_setup
    eng = hip_tiny["f32"]
    D, seq_n, n_seq = 128, 70, 3
    M = seq_n * n_seq
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, D, generator=g).to(dtype)
    W = (torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(dtype)
    b = torch.randn(3 * D, generator=g) * 0.1
    ropes = orc.rope_tables(seq_n)
    y = A.float() @ W.float().t() + b
    q, k, v = y.split(D, dim=-1)
    pos_tab = [t.repeat(n_seq, 1) for t in ropes]
    qr = Oracle.rope_apply(q.reshape(M, 2, 64), pos_tab[0], pos_tab[1]).reshape(M, D)
    kr = Oracle.rope_apply(k.reshape(M, 2, 64), pos_tab[2], pos_tab[3]).reshape(M, D)
    ref = torch.cat([qr, kr, v], dim=-1)
    dr = [t.contiguous().to(gu.DEV) for t in ropes]
    got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=1, ropes=dr, seq_n=seq_n, rope_dim=D)
    assert gu.rel_err(got, ref) < _tol(dtype)
    # compact (cos, sin) pair tables + the persistent 256-tile kernel (bf16) must give the same rotation
    cs = [torch.zeros(seq_n, 64, device=gu.DEV) for _ in range(2)]
    for i in range(2):
        gu.check(eng, eng.lib.vv_rope_compact(eng.ctx, dr[2 * i].data_ptr(), dr[2 * i + 1].data_ptr(), cs[i].data_ptr(), seq_n, gu.stream()))
    got2 = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=1, ropes=dr + cs, seq_n=seq_n, rope_dim=D, tile=128)
    assert gu.rel_err(got2, ref) < _tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(700, 512, 192), (256, 256, 64), (5000, 256, 128)])
def test_gemm_big_tile(hip_tiny, dtype, M, N, K):
    """The 256x256 / 8-wave variant (auto-selected for M >= 4096) on forced and ragged shapes, all epilogues."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b, gate = torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g)
    y = A.float() @ W.float().t() + b
    got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), act=1, tile=256)
    assert gu.rel_err(got, F.gelu(y, approximate="tanh")) < _tol(dtype)
    small = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), act=1, tile=128)
    assert gu.rel_err(got, small.float()) < (8e-3 if dtype == torch.bfloat16 else 1e-6)   # tilings agree to rounding (bias enters first in the persistent kernel)
    x0 = torch.randn(M, N, generator=g)
    got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=2, out_dtype=rt.VV_F32, gate=gate.to(gu.DEV),
                  C_io=x0.clone().to(gu.DEV), tile=256)
    assert gu.rel_err(got, x0 + gate * y) < (2e-3 if dtype == torch.bfloat16 else TOL_F32)
    dl = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=3, gate=gate.to(gu.DEV), tile=256)
    assert gu.rel_err(dl, gate * y) < _tol(dtype)
    assert gu.rel_err(dl, gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=3, gate=gate.to(gu.DEV), tile=128).float()) < (8e-3 if dtype == torch.bfloat16 else 1e-6)


# ------------------------------------------------------------------------------------ packed ragged rows
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packed_rows_rope_attention_posconv(hip_tiny, dtype, tiny_setup):
    """Ragged sequences stored back to back (row_start / rope_pos): the QKV rope epilogue, attention and the conv position
    embedding must give, per sequence, what the padded layout gives -- and must not touch a neighbour's rows."""
    rt, gu = _imports()
    from oracle.vv_oracle import Oracle
    spec, _, orc = tiny_setup
    eng = hip_tiny["f32"]
    lens = [150, 1, 97, 129, 300]
    D, heads, G, KW = 128, 2, 2, 31
    starts = [0]
    for L in lens:
        starts.append(starts[-1] + L)
    R, N = starts[-1], max(lens)
    g = torch.Generator().manual_seed(23)
    rs = torch.tensor(starts[:-1], dtype=torch.int32, device=gu.DEV)
    sl = torch.tensor(lens, dtype=torch.int32, device=gu.DEV)
    pos = torch.cat([torch.arange(L) for L in lens]).to(torch.int32)
    tol = _tol(dtype)
    # --- QKV GEMM + rope at per-row positions (forced 256 tile exercises the persistent kernel's lookup in bf16)
    Dg = 256                                                       # 3 * Dg is a multiple of the 256 tile
    A = torch.randn(R, Dg, generator=g).to(dtype)
    W = (torch.randn(3 * Dg, Dg, generator=g) / math.sqrt(Dg)).to(dtype)
    b = torch.randn(3 * Dg, generator=g) * 0.1
    ropes = orc.rope_tables(N)
    y = A.float() @ W.float().t() + b
    q, k, v = y.split(Dg, dim=-1)
    tabs = [t[pos.long()] for t in ropes]
    ref = torch.cat([Oracle.rope_apply(q.reshape(R, Dg // 64, 64), tabs[0], tabs[1]).reshape(R, Dg),
                     Oracle.rope_apply(k.reshape(R, Dg // 64, 64), tabs[2], tabs[3]).reshape(R, Dg), v], dim=-1)
    dr = [t.contiguous().to(gu.DEV) for t in ropes]
    cs = [torch.zeros(N, 64, device=gu.DEV) for _ in range(2)]
    for i in range(2):
        gu.check(eng, eng.lib.vv_rope_compact(eng.ctx, dr[2 * i].data_ptr(), dr[2 * i + 1].data_ptr(), cs[i].data_ptr(), N, gu.stream()))
    for tile in (128, 256):
        got = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=1, ropes=dr + cs, seq_n=N, rope_dim=Dg, tile=tile, rope_pos=pos.to(gu.DEV))
        assert gu.rel_err(got, ref) < tol, tile
    # row-gathered compact tables (vv_rope_rows, what vv_transformer_steps builds once per call): the persistent kernel then
    # needs no position lookup; same values gathered, so the result is bit-identical to the position-table path
    rows = [torch.zeros(R, 64, device=gu.DEV) for _ in range(2)]
    for i in range(2):
        gu.check(eng, eng.lib.vv_rope_rows(eng.ctx, cs[i].data_ptr(), pos.to(gu.DEV).data_ptr(), rows[i].data_ptr(), R, gu.stream()))
    torch.cuda.synchronize()
    assert torch.equal(rows[0], cs[0][pos.long().to(gu.DEV)]) and torch.equal(rows[1], cs[1][pos.long().to(gu.DEV)])
    got_rows = gu.gemm(eng, A.to(gu.DEV), W.to(gu.DEV), bias=b.to(gu.DEV), mode=1, ropes=dr + rows, seq_n=N, rope_dim=Dg, tile=256,
                       rope_pos=pos.to(gu.DEV), rope_by_row=1)
    assert torch.equal(got_rows, got) if dtype == torch.bfloat16 else gu.rel_err(got_rows, ref) < tol
    # --- attention over packed rows; a sentinel row after each sequence region checks nothing spills
    qkv = torch.randn(R, 3 * D, generator=g)
    qkv[:, :D] *= 0.35
    qkv = qkv.to(dtype)
    out = torch.full((R + 8, D), 7.0, dtype=dtype, device=gu.DEV)
    a = rt.vv_attn_args()
    a.dtype = rt.VV_BF16 if dtype == torch.bfloat16 else rt.VV_F32
    dq = qkv.to(gu.DEV)
    a.qkv, a.ld_qkv, a.out, a.ld_out = dq.data_ptr(), 3 * D, out.data_ptr(), D
    a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len, a.row_start = len(lens), N, heads, D, sl.data_ptr(), rs.data_ptr()
    assert eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()) != 0            # packed rows without the buffer's row count are refused
    a.total_rows = R                          # the last tile of the last sequence reads past row R: the resource bound makes that zeros
    gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))
    torch.cuda.synchronize()
    f = qkv.float().reshape(R, 3, heads, 64)
    for s_, L in enumerate(lens):
        blk = f[starts[s_]: starts[s_] + L]
        sc = torch.einsum("qhd,khd->hqk", blk[:, 0], blk[:, 1])
        want = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), blk[:, 2]).reshape(L, D)
        assert gu.rel_err(out[starts[s_]: starts[s_] + L], want) < tol, s_
    assert bool((out[R:].float() == 7.0).all())
    a.kv_len = None
    assert eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()) != 0            # packed rows without lengths are refused
    # --- conv position embedding over packed rows
    x = torch.randn(R, D, generator=g).to(dtype)
    w = (torch.randn(D, 64, KW, generator=g) / math.sqrt(64 * KW)).to(dtype)
    bb = torch.randn(D, generator=g) * 0.1
    wp = w.reshape(G, 64, 64, KW).permute(0, 3, 1, 2).contiguous() if dtype == torch.bfloat16 else w.reshape(G, 64, 64, KW).permute(0, 3, 2, 1).contiguous()
    pout = torch.full((R + 8, D), 7.0, dtype=torch.float32, device=gu.DEV)
    dx, dw, db = x.to(gu.DEV), wp.to(gu.DEV), bb.to(gu.DEV)
    pa = rt.vv_posconv_args()
    pa.dtype, pa.out_dtype = (rt.VV_BF16 if dtype == torch.bfloat16 else rt.VV_F32), rt.VV_F32
    pa.in_, pa.ld_in, pa.W, pa.bias, pa.out, pa.ld_out = dx.data_ptr(), D, dw.data_ptr(), db.data_ptr(), pout.data_ptr(), D
    pa.n_seq, pa.seq_n, pa.groups, pa.KW, pa.B, pa.seq_len, pa.row_start = len(lens), N, G, KW, len(lens), sl.data_ptr(), rs.data_ptr()
    gu.check(eng, eng.lib.vv_posconv(eng.ctx, C.byref(pa), gu.stream()))
    torch.cuda.synchronize()
    for s_, L in enumerate(lens):
        xs = x.float()[starts[s_]: starts[s_] + L]
        want = F.mish(F.conv1d(xs.t().unsqueeze(0), w.float(), bb, padding=KW // 2, groups=G)).squeeze(0).t()
        assert gu.rel_err(pout[starts[s_]: starts[s_] + L], want) < (5e-3 if dtype == torch.bfloat16 else TOL_F32), s_
    assert bool((pout[R:] == 7.0).all())


@pytest.mark.parametrize("N,K,act", [(1024, 1024, 0), (2048, 512, 1), (768, 256, 1)])
def test_gemm_persistent_blocks_walk_several_tiles(hip_tiny, N, K, act):
    """More tiles than CUs: every persistent workgroup runs 2+ tiles back to back, which is the only path where the first
    K-tile of a tile overlaps the previous tile's 16 epilogue stores (vmcnt(24) instead of vmcnt(8) in the store mode) and
    where the staging schedule rolls over from one tile into the next.  Checked against an fp32 matmul of the same bf16
    operands on the device; M is ragged so the last m-tile is partial (no relaxed wait after a partial store)."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    M = 256 * 90 + 77
    g = torch.Generator().manual_seed(N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(gu.DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(gu.DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(gu.DEV)
    gate = torch.randn(N, generator=g).to(gu.DEV)
    y = A.float() @ W.float().t() + b
    want = F.gelu(y, approximate="tanh") if act else y
    for rep in range(3):                       # timing-dependent hazards would not show on every launch
        got = gu.gemm(eng, A, W, bias=b, act=act, tile=256)
        assert gu.rel_err(got, want) < TOL_BF16, rep
    dl = gu.gemm(eng, A, W, bias=b, mode=3, gate=gate, tile=256)
    assert gu.rel_err(dl, gate * y) < TOL_BF16
    if N == 768:                               # rope epilogue over several tiles per workgroup (positions = row % seq_n)
        from oracle.vv_oracle import Oracle
        seq_n, Dq = 1600, 256
        pos = torch.arange(seq_n, dtype=torch.float32)
        inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
        ang = torch.repeat_interleave(pos[:, None] * inv[None, :], 2, dim=1)          # pair-duplicated tables [seq_n][64]
        ropes = [ang.cos() * 0.125, ang.sin() * 0.125, ang.cos(), ang.sin()]
        q, k, v = y.cpu().split(Dq, dim=-1)
        idx = torch.arange(M) % seq_n
        tabs = [t[idx] for t in ropes]
        ref = torch.cat([Oracle.rope_apply(q.reshape(M, Dq // 64, 64), tabs[0], tabs[1]).reshape(M, Dq),
                         Oracle.rope_apply(k.reshape(M, Dq // 64, 64), tabs[2], tabs[3]).reshape(M, Dq), v], dim=-1)
        dr = [t.contiguous().to(gu.DEV) for t in ropes]
        got = gu.gemm(eng, A, W, bias=b, mode=1, ropes=dr, seq_n=seq_n, rope_dim=Dq, tile=256)
        assert gu.rel_err(got, ref) < TOL_BF16


def _tail_sum(Ct):
    """What vv_layernorm makes of a split-K tail: the fp32 K parts summed in part order, the sum rounded to bf16 once."""
    acc = Ct[0].clone()
    for p_ in range(1, Ct.shape[0]):
        acc = acc + Ct[p_]
    return acc.to(torch.bfloat16)


def test_gemm_headline_shape_every_row(hip_tiny):
    """The four GEMMs of a DiT block at the HEADLINE size (M = 102,400 rows = 32 utterances x 1,600 frames x 2 branches; dim 1024, FF 2048),
    EVERY output element against an fp32 product of the same bf16 operands formed on the device: 400 row panels x 4..12 n-tiles over 256
    persistent workgroups (6.25 - 18.75 tiles each), the split-K tail of the two N = 1024 GEMMs as bench.py runs them, row-gathered
    rope tables as vv_transformer_steps builds them.  Size-specific faults (a wrong wave tile somewhere in the 1,600..4,800 tiles of a
    launch) do not show in the small-shape tests above; three launches each, since a timing-dependent one need not show every time."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    dev = gu.DEV
    M, D, FF, seq_n = 102400, 1024, 2048, 1600
    g = torch.Generator().manual_seed(102400)
    x = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    hid = (torch.randn(M, FF, generator=g) * 0.5).to(torch.bfloat16).to(dev)

    def wgt(n, k):
        return (torch.randn(n, k, generator=g) / math.sqrt(k)).to(torch.bfloat16).to(dev), (torch.randn(n, generator=g) * 0.1).to(dev)

    def check_rows(got_fn, want_fn, scale):
        worst = 0.0
        for lo in range(0, M, 12800):
            worst = max(worst, float((got_fn(lo, lo + 12800).float() - want_fn(lo, lo + 12800)).abs().max()))
        assert worst < TOL_BF16 * scale, (worst, scale)

    # ---- QKV + rope (row-gathered compact tables, positions = row % seq_n)
    Wq, bq = wgt(3 * D, D)
    pos = (torch.arange(M, dtype=torch.int32) % seq_n).to(dev)
    ang = torch.repeat_interleave(torch.arange(seq_n, dtype=torch.float32)[:, None] / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))[None, :], 2, dim=1)
    ropes = [t.contiguous().to(dev) for t in (ang.cos() * 0.125, ang.sin() * 0.125, ang.cos(), ang.sin())]
    cs = [torch.zeros(seq_n, 64, device=dev) for _ in range(2)]
    rows = [torch.zeros(M, 64, device=dev) for _ in range(2)]
    for i in range(2):
        gu.check(eng, eng.lib.vv_rope_compact(eng.ctx, ropes[2 * i].data_ptr(), ropes[2 * i + 1].data_ptr(), cs[i].data_ptr(), seq_n, gu.stream()))
        gu.check(eng, eng.lib.vv_rope_rows(eng.ctx, cs[i].data_ptr(), pos.data_ptr(), rows[i].data_ptr(), M, gu.stream()))

    def want_qkv(lo, hi):
        y = x[lo:hi].float() @ Wq.float().t() + bq
        p = pos[lo:hi].long()
        out = y.clone()
        for part in (0, 1):
            c, sn = ropes[2 * part][p], ropes[2 * part + 1][p]                       # [rows][64] pair-duplicated
            v = y[:, part * D:(part + 1) * D].reshape(hi - lo, D // 64, 32, 2)
            cc, ss = c[:, 0::2].reshape(hi - lo, 1, 32), sn[:, 0::2].reshape(hi - lo, 1, 32)
            out[:, part * D:(part + 1) * D] = torch.stack((v[..., 0] * cc - v[..., 1] * ss, v[..., 1] * cc + v[..., 0] * ss), dim=-1).reshape(hi - lo, D)
        return out
    ref_scale = float((x[:4096].float() @ Wq.float().t() + bq).abs().max())
    for rep in range(3):
        got = gu.gemm(eng, x, Wq, bias=bq, mode=1, ropes=ropes + rows, seq_n=seq_n, rope_dim=D, rope_pos=pos, rope_by_row=1)
        check_rows(lambda lo, hi: got[lo:hi], want_qkv, ref_scale)
    del got

    # ---- FF1 + tanh-GELU
    W1, b1 = wgt(FF, D)
    s1 = float(F.gelu(x[:4096].float() @ W1.float().t() + b1, approximate="tanh").abs().max())
    for rep in range(3):
        got = gu.gemm(eng, x, W1, bias=b1, act=1)
        check_rows(lambda lo, hi: got[lo:hi], lambda lo, hi: F.gelu(x[lo:hi].float() @ W1.float().t() + b1, approximate="tanh"), s1)
    del got

    # ---- out-projection (K = 1024) and FF2 (K = 2048): gate-store with the split-K tail planned for this device, parts summed as the
    # consuming LayerNorm does
    for A, K in ((x, D), (hid, FF)):
        Wo, bo = wgt(D, K)
        gate = torch.randn(D, generator=g).to(dev)
        row0, parts = gu.gemm_tail_plan(eng, M, D, K)
        so = float((gate * (A[:4096].float() @ Wo.float().t() + bo)).abs().max())
        for rep in range(3):
            if parts:
                Ct = torch.full((parts, M - row0, D), 7.0, dtype=torch.float32, device=dev)
                got = gu.gemm(eng, A, Wo, bias=bo, mode=3, gate=gate, tail=(Ct, row0, parts))
                tail_sum = _tail_sum(Ct)                  # fp32 K parts in part order, rounded once: the delta the LayerNorm forms

                def got_fn(lo, hi, got=got, tail_sum=tail_sum, row0=row0):
                    o = got[lo:hi].float()
                    if hi > row0:
                        a0 = max(lo, row0)
                        o[a0 - lo:] = tail_sum[a0 - row0:hi - row0].float()      # the tail rows of C are not written
                    return o
            else:
                got = gu.gemm(eng, A, Wo, bias=bo, mode=3, gate=gate)
                got_fn = lambda lo, hi, got=got: got[lo:hi]
            check_rows(got_fn, lambda lo, hi: gate * (A[lo:hi].float() @ Wo.float().t() + bo), so)


def test_gemm_rejects_bad_shapes(hip_tiny):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    A = torch.zeros(8, 64, device=gu.DEV)
    W = torch.zeros(100, 64, device=gu.DEV)     # N not a multiple of 128
    a = rt.vv_gemm_args()
    a.dtype = a.out_dtype = rt.VV_F32
    out = torch.zeros(8, 100, device=gu.DEV)
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), 64, W.data_ptr(), 64, out.data_ptr(), 100, 8, 100, 64
    rc = eng.lib.vv_gemm(eng.ctx, C.byref(a), gu.stream())
    assert rc != 0 and b"multiple of 128" in eng.lib.vv_last_error(eng.ctx)


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seq_n,lens", [(200, [200, 137, 64]), (64, [64, 1, 33]), (333, [333, 300, 129])])
def test_attention(hip_tiny, dtype, seq_n, lens):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    heads, D, n_seq = 2, 128, len(lens)
    g = torch.Generator().manual_seed(seq_n)
    qkv = torch.randn(n_seq * seq_n, 3 * D, generator=g)
    qkv[:, :D] *= 0.35          # q carries the softmax scale in the real pipeline
    qkv = qkv.to(dtype)
    out = torch.zeros(n_seq * seq_n, D, dtype=dtype, device=gu.DEV)
    kv = torch.tensor(lens, dtype=torch.int32, device=gu.DEV)
    a = rt.vv_attn_args()
    a.dtype = rt.VV_BF16 if dtype == torch.bfloat16 else rt.VV_F32
    dq = qkv.to(gu.DEV)
    a.qkv, a.ld_qkv, a.out, a.ld_out = dq.data_ptr(), 3 * D, out.data_ptr(), D
    a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len = n_seq, seq_n, heads, D, kv.data_ptr()
    gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))
    torch.cuda.synchronize()
    f = qkv.float().reshape(n_seq, seq_n, 3, heads, 64)
    for s, L in enumerate(lens):
        q, k, v = f[s, :, 0], f[s, :L, 1], f[s, :L, 2]
        sc = torch.einsum("qhd,khd->hqk", q, k)
        ref = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), v).reshape(seq_n, D)
        got = out[s * seq_n:(s + 1) * seq_n]
        assert gu.rel_err(got[:L], ref[:L]) < _tol(dtype), (s, gu.rel_err(got[:L], ref[:L]))


def test_attention_headline_shape_every_row(hip_tiny):
    """bf16 attention at the headline size (64 sequences = 32 utterances x 2 branches, 1,600 frames, 16 heads of 64): every output row
    of every sequence against an fp32 softmax(QK^T)V of the same bf16 operands formed on the device, uniform lengths (what bench.py runs)
    and ragged key lengths (masked tails, speculative-softmax redo on the partial last tile)."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    dev = gu.DEV
    n_seq, seq_n, heads, D = 64, 1600, 16, 1024
    g = torch.Generator().manual_seed(6416)
    qkv = torch.randn(n_seq * seq_n, 3 * D, generator=g)
    qkv[:, :D] *= 0.35                                                  # q carries the 1/sqrt(64) of the rope tables in the path
    qkv = qkv.to(torch.bfloat16).to(dev)
    lens_ragged = [seq_n - (37 * s_) % 900 for s_ in range(n_seq)]
    for lens in ([seq_n] * n_seq, lens_ragged):
        out = torch.zeros(n_seq * seq_n, D, dtype=torch.bfloat16, device=dev)
        kv = torch.tensor(lens, dtype=torch.int32, device=dev)
        a = rt.vv_attn_args()
        a.dtype = rt.VV_BF16
        a.qkv, a.ld_qkv, a.out, a.ld_out = qkv.data_ptr(), 3 * D, out.data_ptr(), D
        a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len = n_seq, seq_n, heads, D, kv.data_ptr()
        gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        worst = 0.0
        for s_, L in enumerate(lens):
            f = qkv[s_ * seq_n:(s_ + 1) * seq_n].float().reshape(seq_n, 3, heads, 64)
            sc = torch.einsum("qhd,khd->hqk", f[:, 0], f[:L, 1])
            ref = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), f[:L, 2]).reshape(seq_n, D)
            got = out[s_ * seq_n:(s_ + 1) * seq_n].float()
            worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
        assert worst < TOL_BF16, (worst, lens[:4])


@pytest.mark.parametrize("M_seq,tile", [(3, 128), (24, 256)])
def test_query_rope_in_attention_equals_rope_in_the_gemm(hip_tiny, M_seq, tile):
    """Round 4: the query side of the rope moved from the QKV GEMM's epilogue (rope_skip_q) into the attention kernel's Q load
    (vv_attn_args.rope_cs_q; position = row inside the sequence; ragged key lengths; the compact q table carries the softmax scale).
    Route A = everything in the GEMM (the round-3 path, still the fp32 model's), route B = the split.  Both against an fp32 attention
    over fp32-roped projections of the same bf16 operands, and against each other, in the bf16 tolerance class -- with the 128 x 128
    kernel (3 sequences) and the persistent 256 x 256 kernel (24 sequences = 6,960 rows)."""
    rt, gu = _imports()
    from oracle.vv_oracle import Oracle
    eng = hip_tiny["f32"]
    dev = gu.DEV
    seq_n, heads = 290, 4
    D = heads * 64
    lens = [seq_n - (41 * i) % 200 for i in range(M_seq)]
    M = M_seq * seq_n
    g = torch.Generator().manual_seed(290 + M_seq)
    A = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(torch.bfloat16).to(dev)
    b = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    ang = torch.repeat_interleave(torch.arange(seq_n, dtype=torch.float32)[:, None] * inv[None, :], 2, dim=1)
    ropes = [t.contiguous().to(dev) for t in (ang.cos() * 0.125, ang.sin() * 0.125, ang.cos(), ang.sin())]
    cs_q = torch.zeros(seq_n, 64, device=dev)
    gu.check(eng, eng.lib.vv_rope_compact(eng.ctx, ropes[0].data_ptr(), ropes[1].data_ptr(), cs_q.data_ptr(), seq_n, gu.stream()))
    kv = torch.tensor(lens, dtype=torch.int32, device=dev)

    def attention(qkv, table):
        out = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
        a = rt.vv_attn_args()
        a.dtype = rt.VV_BF16
        a.qkv, a.ld_qkv, a.out, a.ld_out = qkv.data_ptr(), 3 * D, out.data_ptr(), D
        a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len = M_seq, seq_n, heads, D, kv.data_ptr()
        a.rope_cs_q = None if table is None else table.data_ptr()
        gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        return out

    qkv_a = gu.gemm(eng, A, W, bias=b, mode=1, ropes=ropes, seq_n=seq_n, rope_dim=D, tile=tile)
    qkv_b = gu.gemm(eng, A, W, bias=b, mode=1, ropes=ropes, seq_n=seq_n, rope_dim=D, tile=tile, rope_skip_q=1)
    assert torch.equal(qkv_a[:, D:], qkv_b[:, D:])                        # k (roped) and v columns are untouched by the option
    plain = (A.float() @ W.float().t() + b)
    assert gu.rel_err(qkv_b[:, :D], plain[:, :D]) < TOL_BF16             # q columns: plain projection, no rope
    out_a, out_b = attention(qkv_a, None), attention(qkv_b, cs_q)
    idx = (torch.arange(M) % seq_n)
    tabs = [t.cpu()[idx] for t in ropes]
    y = plain.cpu()
    q = Oracle.rope_apply(y[:, :D].reshape(M, heads, 64), tabs[0], tabs[1]).reshape(M_seq, seq_n, heads, 64)
    k = Oracle.rope_apply(y[:, D:2 * D].reshape(M, heads, 64), tabs[2], tabs[3]).reshape(M_seq, seq_n, heads, 64)
    v = y[:, 2 * D:].reshape(M_seq, seq_n, heads, 64)
    worst_a = worst_b = worst_ab = 0.0
    for s_, L in enumerate(lens):
        sc = torch.einsum("qhd,khd->hqk", q[s_], k[s_, :L])
        ref = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), v[s_, :L]).reshape(seq_n, D)[:L]
        ga, gb = out_a[s_ * seq_n: s_ * seq_n + L].float().cpu(), out_b[s_ * seq_n: s_ * seq_n + L].float().cpu()
        sc_ = float(ref.abs().max())
        worst_a, worst_b = max(worst_a, float((ga - ref).abs().max()) / sc_), max(worst_b, float((gb - ref).abs().max()) / sc_)
        worst_ab = max(worst_ab, float((ga - gb).abs().max()) / sc_)
    print(f"\n[q rope in attention, {M_seq} sequences] vs fp32: rope in the GEMM {worst_a:.2e}, rope in attention {worst_b:.2e}; A vs B {worst_ab:.2e}")
    assert worst_a < TOL_BF16 and worst_b < TOL_BF16 and worst_ab < TOL_BF16
    with pytest.raises(AssertionError, match="bf16 kernel only"):
        a = rt.vv_attn_args()
        f32 = torch.zeros(seq_n, 3 * D, device=dev)
        o32 = torch.zeros(seq_n, D, device=dev)
        a.dtype, a.qkv, a.ld_qkv, a.out, a.ld_out, a.n_seq, a.seq_n, a.heads, a.dim, a.rope_cs_q = rt.VV_F32, f32.data_ptr(), 3 * D, o32.data_ptr(), D, 1, seq_n, heads, D, cs_q.data_ptr()
        gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))


def test_computed_rope_matches_the_tables(hip_tiny):
    """Round 4: with the standard tables declared (vv_gemm_args.rope_theta / vv_set_rope_theta) the bf16 QKV epilogue COMPUTES cos / sin
    of the q and k columns from the position (v_exp / v_fract / v_cos / v_sin: no table load behind the stores) and leaves the softmax
    scale to the attention kernel (vv_attn_args.q_scale).  Against an fp32 projection roped with float64-defined tables: bf16 tolerance at
    positions up to 4,095 (where an fp32 angle is ~3e-4 rad off); the 128 x 128 and the persistent kernel give IDENTICAL bits (one
    formula: an item in a batch equals the item alone); per-row position tables (packed rows) and the table-free uniform form agree;
    attention with q_scale equals attention on a q that carried the scale."""
    rt, gu = _imports()
    from oracle.vv_oracle import Oracle
    eng = hip_tiny["f32"]
    dev = gu.DEV
    seq_n, heads, theta = 4096, 2, 10000.0
    D = heads * 64
    M = 2 * seq_n
    g = torch.Generator().manual_seed(4096)
    A = torch.randn(M, 256, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(3 * D + 128, 256, generator=g) / 16.0).to(torch.bfloat16).to(dev)          # N = 512: a multiple of the 256 tile
    b = (torch.randn(3 * D + 128, generator=g) * 0.1).to(dev)
    inv = 1.0 / (theta ** (torch.arange(0, 64, 2, dtype=torch.float64) / 64))
    ang = torch.repeat_interleave(torch.arange(seq_n, dtype=torch.float64)[:, None] * inv[None, :], 2, dim=1)
    cos, sin = ang.cos().float(), ang.sin().float()
    ropes = [t.contiguous().to(dev) for t in (cos * 0.125, sin * 0.125, cos, sin)]
    pos = (torch.arange(M, dtype=torch.int32) % seq_n).to(dev)
    y = (A.float() @ W.float().t() + b).cpu()
    idx = torch.arange(M) % seq_n
    want = y.clone()
    want[:, :D] = Oracle.rope_apply(y[:, :D].reshape(M, heads, 64), cos[idx], sin[idx]).reshape(M, D)            # q: NO scale
    want[:, D:2 * D] = Oracle.rope_apply(y[:, D:2 * D].reshape(M, heads, 64), cos[idx], sin[idx]).reshape(M, D)
    outs = {}
    for tile in (128, 256):
        for with_pos in (False, True):
            got = gu.gemm(eng, A, W, bias=b, mode=1, ropes=ropes, seq_n=seq_n, rope_dim=D, tile=tile, rope_theta=theta,
                          rope_pos=pos if with_pos else None)
            outs[(tile, with_pos)] = got
            err = gu.rel_err(got, want)
            hi = float((got[-64:, :2 * D].float().cpu() - want[-64:, :2 * D]).abs().max() / want[:, :2 * D].abs().max())
            print(f"\n[computed rope, tile {tile}, pos table {with_pos}] rel err {err:.2e}; rows at positions 4032..4095: {hi:.2e}")
            assert err < TOL_BF16
    assert torch.equal(outs[(128, False)], outs[(256, False)]) and torch.equal(outs[(128, True)], outs[(256, True)])
    assert torch.equal(outs[(256, False)], outs[(256, True)])
    tab = gu.gemm(eng, A, W, bias=b, mode=1, ropes=ropes, seq_n=seq_n, rope_dim=D, tile=256)          # the table form (q carries 0.125)
    dk = float((tab[:, D:2 * D].float() - outs[(256, False)][:, D:2 * D].float()).abs().max() / want[:, D:2 * D].abs().max())
    print(f"[computed rope] k columns, computed vs table form: max diff {dk:.2e} of the range (bf16 ulp = 3.9e-3)")
    assert dk < 8e-3
    # attention: q_scale on an unscaled q == the same q pre-scaled by an exact power of two
    qkv = outs[(256, False)][:seq_n, : 3 * D].contiguous()
    kv = torch.tensor([seq_n], dtype=torch.int32, device=dev)

    def attention(x, q_scale):
        out = torch.zeros(seq_n, D, dtype=torch.bfloat16, device=dev)
        a = rt.vv_attn_args()
        a.dtype, a.qkv, a.ld_qkv, a.out, a.ld_out = rt.VV_BF16, x.data_ptr(), 3 * D, out.data_ptr(), D
        a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len, a.q_scale = 1, seq_n, heads, D, kv.data_ptr(), q_scale
        gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        return out
    pre = qkv.clone()
    pre[:, :D] = (pre[:, :D].float() * 0.125).to(torch.bfloat16)                 # exact in bf16
    assert torch.equal(attention(qkv, 0.125), attention(pre, 0.0))


@pytest.mark.parametrize("M", [1600, 3200 + 77, 4800])
def test_gemm_three_tilings_give_the_same_bits(hip_tiny, M):
    """Round 4: vv_gemm picks among three bf16 tilings by launch size -- 64-token x 128-feature tiles (launches that do not fill the chip),
    128 x 128, and the persistent 256 x 256 kernel -- and a row's result must not depend on the choice (an item alone and the same item
    in a batch take different ones).  Round 5: a fourth, 64 x 64 tiles on a three-stage LDS ring with a counted wait (tile code 6464;
    K = 512 here is 8 K-tiles: ring start, steady state and drain; K = 64 and 128 -- one and two K-tiles -- in the case below).  All three contract K in the same order with the same MFMA from bias-started accumulators and share
    the epilogue arithmetic: plain store, tanh-GELU store, gate store and the computed-rope QKV epilogue are EQUAL arrays, ragged M
    included (the last tile of each tiling is partial in a different place); and equal to what the automatic choice produces."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    dev = gu.DEV
    g = torch.Generator().manual_seed(M)
    K, N = 512, 1024
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    gate = torch.randn(N, generator=g).to(dev)
    dummy = torch.zeros(1600, 64, device=dev)
    ropes = [dummy, dummy, dummy, dummy]
    forms = {"store": dict(), "gelu": dict(act=1), "gate": dict(mode=3, gate=gate),
             "rope": dict(mode=1, ropes=ropes, seq_n=800, rope_dim=512, rope_theta=10000.0)}
    y = A.float() @ W.float().t() + b
    for name, kw in forms.items():
        outs = {t: gu.gemm(eng, A, W, bias=b, tile=t, **kw) for t in (64, 128, 256, 0, 6464)}
        outs["shared"] = gu.gemm(eng, A, W, bias=b, tile=0, chip_share=2, **kw)      # the automatic choice priced for half of the chip
        for t in (128, 256, 0, "shared", 6464):
            assert torch.equal(outs[t], outs[64]), (name, t, float((outs[t].float() - outs[64].float()).abs().max()))
        if name == "store":
            assert gu.rel_err(outs[64], y) < TOL_BF16
        if name == "gate":
            assert gu.rel_err(outs[64], gate * y) < TOL_BF16


@pytest.mark.parametrize("K", [64, 128, 192, 2048])
def test_gemm_ring_tiling_short_and_long_k(hip_tiny, K):
    """The three-stage ring of the 64 x 64 tiling at the ends of its range: one K-tile (nothing to prefetch), two (no steady state), three,
    and the FF2 depth (32 K-tiles), ragged M, gate store: equal to the 128 x 128 kernel bit for bit, fresh rows past M untouched."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(K)
    M, N = 1600 + 37, 1024
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(gu.DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(gu.DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(gu.DEV)
    gate = torch.randn(N, generator=g).to(gu.DEV)
    for kw in (dict(), dict(mode=3, gate=gate)):
        ref = gu.gemm(eng, A, W, bias=b, tile=128, **kw)
        for rep in range(3):                                   # the ring is a race if its counts are wrong: more than one launch
            for t in (6464,):
                got = gu.gemm(eng, A, W, bias=b, tile=t, **kw)
                assert torch.equal(got, ref), (K, t, kw.keys(), rep, float((got.float() - ref.float()).abs().max()))


def test_attention_spiked_max(hip_tiny):
    """Online-softmax rescale branch: a key late in the sequence dominates one query row."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    seq_n, D = 256, 128
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(seq_n, 3 * D, generator=g) * 0.3
    qkv[5, :64] = 3.0
    qkv[200, D:D + 64] = 3.0      # q5 . k200 = 576 >> everything else: 2^831 overflows -> the careful path centres the row
    # round 4 (no reference until a row needs one): a row whose FIRST tile underflows entirely (q7 . k_j = -576 for every key of tile 0 but
    # one) must take the careful path there and centre DOWN, then move up again when the ordinary keys of the later tiles arrive
    qkv[7, :64] = 3.0
    qkv[:64, D:D + 64] = -3.0
    qkv[20, D:D + 64] = torch.randn(64, generator=g) * 0.3
    for dtype in (torch.float32, torch.bfloat16):
        x = qkv.to(dtype)
        out = torch.zeros(seq_n, D, dtype=dtype, device=gu.DEV)
        a = rt.vv_attn_args()
        a.dtype = rt.VV_BF16 if dtype == torch.bfloat16 else rt.VV_F32
        dq = x.to(gu.DEV)
        a.qkv, a.ld_qkv, a.out, a.ld_out, a.n_seq, a.seq_n, a.heads, a.dim, a.kv_len = dq.data_ptr(), 3 * D, out.data_ptr(), D, 1, seq_n, 2, D, None
        gu.check(eng, eng.lib.vv_attention(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        f = x.float().reshape(seq_n, 3, 2, 64)
        sc = torch.einsum("qhd,khd->hqk", f[:, 0], f[:, 1])
        ref = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), f[:, 2]).reshape(seq_n, D)
        assert gu.rel_err(out, ref) < _tol(dtype)


# ------------------------------------------------------------------------------------ LayerNorm / AdaLN
@pytest.mark.parametrize("D", [128, 512, 1024])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_layernorm_modulate(hip_tiny, D, out_dtype):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    R = 37
    g = torch.Generator().manual_seed(D)
    x = torch.randn(R, D, generator=g) * 2 + 0.5
    sc, sh = torch.randn(D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.3
    y = torch.zeros(R, D, dtype=out_dtype, device=gu.DEV)
    dx, dsc, dsh = x.to(gu.DEV), sc.to(gu.DEV), sh.to(gu.DEV)
    for add_one in (1, 0):
        a = rt.vv_ln_args()
        a.out_dtype = rt.VV_BF16 if out_dtype == torch.bfloat16 else rt.VV_F32
        a.x, a.ldx, a.y, a.ldy, a.R, a.D, a.w, a.b, a.add_one, a.eps = dx.data_ptr(), D, y.data_ptr(), D, R, D, dsc.data_ptr(), dsh.data_ptr(), add_one, 1e-6
        gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        ref = F.layer_norm(x, (D,), eps=1e-6) * (sc + add_one) + sh
        assert gu.rel_err(y, ref) < (1e-2 if out_dtype == torch.bfloat16 else 1e-5)
    # fused residual add: x += delta (bf16 or fp32), written back, then normalised
    for ddt in (torch.float32, torch.bfloat16):
        dlt = (torch.randn(R, D, generator=g) * 0.5).to(ddt)
        xx, dd = x.clone().to(gu.DEV), dlt.to(gu.DEV)
        a = rt.vv_ln_args()
        a.out_dtype = rt.VV_BF16 if out_dtype == torch.bfloat16 else rt.VV_F32
        a.x, a.ldx, a.y, a.ldy, a.R, a.D, a.w, a.b, a.add_one, a.eps = xx.data_ptr(), D, y.data_ptr(), D, R, D, dsc.data_ptr(), dsh.data_ptr(), 1, 1e-6
        a.delta, a.delta_dtype, a.ld_delta = dd.data_ptr(), (rt.VV_BF16 if ddt == torch.bfloat16 else rt.VV_F32), D
        gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        xn = x + dlt.float()
        assert torch.equal(xx.cpu(), xn)
        assert gu.rel_err(y, F.layer_norm(xn, (D,), eps=1e-6) * (sc + 1) + sh) < (1e-2 if out_dtype == torch.bfloat16 else 1e-5)
        # once-per-block protocol: keep_x normalises x + d1 without touching x; the later call stores (x + d1) + d2
        d2 = (torch.randn(R, D, generator=g) * 0.5).to(ddt)
        xx, dd2 = x.clone().to(gu.DEV), d2.to(gu.DEV)
        a.x, a.keep_x = xx.data_ptr(), 1
        gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        assert torch.equal(xx.cpu(), x)
        assert gu.rel_err(y, F.layer_norm(xn, (D,), eps=1e-6) * (sc + 1) + sh) < (1e-2 if out_dtype == torch.bfloat16 else 1e-5)
        a.delta2, a.keep_x = dd2.data_ptr(), 0
        gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        xn2 = (x + dlt.float()) + d2.float()
        assert torch.equal(xx.cpu(), xn2)
        assert gu.rel_err(y, F.layer_norm(xn2, (D,), eps=1e-6) * (sc + 1) + sh) < (1e-2 if out_dtype == torch.bfloat16 else 1e-5)


def test_layernorm_headline_shape_every_row(hip_tiny):
    """The block's two LayerNorm passes at the headline size (102,400 rows x 1024): the residual stream after the fused adds is EXACT
    (x + d1, then (x + d1) + d2 with both deltas' split-K tail parts, fp32 adds in the kernel's order) on every row, keep_x leaves x
    untouched, and the modulated bf16 output matches an fp32 LayerNorm of that stream on every row."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    dev = gu.DEV
    R, D = 102400, 1024
    g = torch.Generator().manual_seed(10241024)
    x = (torch.randn(R, D, generator=g) * 2 + 0.25).to(dev)
    sc, sh = (torch.randn(D, generator=g) * 0.3).to(dev), (torch.randn(D, generator=g) * 0.3).to(dev)
    d1 = (torch.randn(R, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    d2 = (torch.randn(R, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    row0, parts = gu.gemm_tail_plan(eng, R, D, 1024)
    assert parts >= 2 and 0 < row0 < R, "the headline shape has a split-K tail on a 256-CU device"
    t1 = (torch.randn(parts, R - row0, D, generator=g) * 0.25).to(dev)          # fp32 K parts of the tail rows (the delta's own tail rows are not read)
    t2 = (torch.randn(parts, R - row0, D, generator=g) * 0.25).to(dev)
    d1[row0:] = float("nan")
    d2[row0:] = float("nan")
    y = torch.zeros(R, D, dtype=torch.bfloat16, device=dev)
    xx = x.clone()
    a = rt.vv_ln_args()
    a.out_dtype = rt.VV_BF16
    a.x, a.ldx, a.y, a.ldy, a.R, a.D, a.w, a.b, a.add_one, a.eps = xx.data_ptr(), D, y.data_ptr(), D, R, D, sc.data_ptr(), sh.data_ptr(), 1, 1e-6
    a.delta, a.delta_dtype, a.ld_delta = d1.data_ptr(), rt.VV_BF16, D
    a.tail_row0, a.delta_tail, a.delta_tail_parts = row0, t1.data_ptr(), parts
    a.keep_x = 1
    gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
    torch.cuda.synchronize()
    assert torch.equal(xx, x)
    def with_tail(d, t):                      # the kernel's order: the K parts are summed in fp32, the sum rounded to bf16 once, then added to the stream
        e = d.float()
        e[row0:] = _tail_sum(t).float()
        return e
    s1 = x + with_tail(d1, t1)

    def ln_err(stream):
        worst = 0.0
        for lo in range(0, R, 12800):
            ref = F.layer_norm(stream[lo:lo + 12800], (D,), eps=1e-6) * (sc + 1) + sh
            worst = max(worst, float((y[lo:lo + 12800].float() - ref).abs().max() / ref.abs().max()))
        return worst
    assert ln_err(s1) < 1e-2
    a.keep_x = 0
    a.delta2, a.delta2_tail, a.delta2_tail_parts = d2.data_ptr(), t2.data_ptr(), parts
    gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
    torch.cuda.synchronize()
    s2 = s1 + with_tail(d2, t2)
    assert torch.equal(xx, s2), float((xx - s2).abs().max())
    assert ln_err(s2) < 1e-2


@pytest.mark.parametrize("N,K,m_tiles,ragged", [(512, 512, 136, 100), (1024, 1024, 68, 0), (1024, 4096, 80, 255), (512, 256, 136, 0)])
def test_gemm_split_k_tail(hip_tiny, N, K, m_tiles, ragged):
    """Split-K tail of the persistent gate-store GEMM (vv_gemm_tail_plan): rows below row0 are bit-identical to the plain
    launch; for rows >= row0 every K part (part 0 with the bias) lands in the fp32 buffer C_tail and C's tail rows stay
    unwritten; the parts summed in fp32 and rounded to bf16 ONCE equal the plain launch's rows except where the fp32
    summation order flips a bf16 rounding (round 4: a row's result no longer depends on its position in the launch).
    Then the consumer: vv_layernorm with delta_tail forms exactly that rounded sum.
    The last case has too few K-tiles for 4 parts (2 K-tiles each are the minimum) and must take 2."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    M = 256 * m_tiles - ragged
    row0, parts = gu.gemm_tail_plan(eng, M, N, K)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count // 8 * 8
    tiles = m_tiles * (N // 256)
    assert tiles % n_cu != 0 and tiles > n_cu, "the case must leave a partial last round on this device"
    assert parts == (2 if K == 256 else 4) and row0 % 2048 == 0 and 0 < row0 < M
    assert (row0 // 256) * (N // 256) % n_cu == 0 and (m_tiles - row0 // 256) * (N // 256) * parts <= n_cu
    g = torch.Generator().manual_seed(N * 7 + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(gu.DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(gu.DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(gu.DEV)
    gate = torch.randn(N, generator=g).to(gu.DEV)
    plain = gu.gemm(eng, A, W, bias=b, mode=3, gate=gate)
    want = gate * (A.float() @ W.float().t() + b)
    for rep in range(2):
        Ct = torch.full((parts, M - row0, N), 7.0, dtype=torch.float32, device=gu.DEV)
        got = gu.gemm(eng, A, W, bias=b, mode=3, gate=gate, tail=(Ct, row0, parts), c_fill=3.0)
        assert torch.equal(got[:row0], plain[:row0])
        assert bool((got[row0:] == 3.0).all()), "the tail rows of C are left to the consumer"
        total = _tail_sum(Ct)
        assert gu.rel_err(total, want[row0:]) < TOL_BF16
        # position independence: the rounded sum IS the plain launch's row, up to bf16 roundings flipped by the fp32 summation order
        flips = (total != plain[row0:])
        assert float(flips.float().mean()) < 2e-3, float(flips.float().mean())
        assert float((total.float() - plain[row0:].float()).abs().max()) <= 2.0 ** -7 * float(plain[row0:].float().abs().max())     # one bf16 ulp
        assert not bool((Ct == 7.0).all(dim=-1).any()), "every tail row of every part is written"
        # the parts really are K ranges: part p alone equals gate * (A[:, p-th K range] W^T [+ bias for part 0]), to fp32 accuracy
        kq = K // parts
        for p_ in range(parts):
            wp = gate * (A[row0:, p_ * kq:(p_ + 1) * kq].float() @ W[:, p_ * kq:(p_ + 1) * kq].float().t() + (b if p_ == 0 else 0.0))
            assert gu.rel_err(Ct[p_], wp) < 1e-4, p_
    # a request that is not the plan for the shape is refused, not guessed at
    a_bad = (Ct, row0 + 2048, parts)
    with pytest.raises(AssertionError, match="split-K tail"):
        gu.gemm(eng, A, W, bias=b, mode=3, gate=gate, tail=a_bad)
    with pytest.raises(AssertionError, match="split-K tail"):
        gu.gemm(eng, A, W, bias=b, mode=0, tail=(Ct, row0, parts))
    if N != 1024:
        return
    # consumer: LayerNorm adds delta + its tail parts (and a second delta with its own tail) before normalising
    x = (torch.randn(M, N, generator=g) * 2).to(gu.DEV)
    sc, sh = (torch.randn(N, generator=g) * 0.3).to(gu.DEV), (torch.randn(N, generator=g) * 0.3).to(gu.DEV)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=gu.DEV)
    d2 = (torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(gu.DEV)
    d2t = (torch.randn(2, M - row0, N, generator=g) * 0.5).to(gu.DEV)
    for two in (False, True):
        xx = x.clone()
        a = rt.vv_ln_args()
        a.out_dtype = rt.VV_BF16
        a.x, a.ldx, a.y, a.ldy, a.R, a.D, a.w, a.b, a.add_one, a.eps = xx.data_ptr(), N, y.data_ptr(), N, M, N, sc.data_ptr(), sh.data_ptr(), 1, 1e-6
        a.delta, a.delta_dtype, a.ld_delta = got.data_ptr(), rt.VV_BF16, N
        a.tail_row0, a.delta_tail, a.delta_tail_parts = row0, Ct.data_ptr(), parts
        if two:
            a.delta2, a.delta2_tail, a.delta2_tail_parts = d2.data_ptr(), d2t.data_ptr(), 2
        gu.check(eng, eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        dsum = got.float()
        dsum[row0:] = _tail_sum(Ct).float()                    # the kernel's order: parts in fp32, one rounding
        xn = x + dsum
        if two:
            e2 = d2.float()
            e2[row0:] = _tail_sum(d2t).float()
            xn = xn + e2
        assert torch.equal(xx, xn)
        assert gu.rel_err(y, F.layer_norm(xn, (N,), eps=1e-6) * (sc + 1) + sh) < 1e-2
    a.delta = None                                                     # a tail without its delta is a caller bug
    assert eng.lib.vv_layernorm(eng.ctx, C.byref(a), gu.stream()) != 0


# ------------------------------------------------------------------------------------ conv position embedding
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_posconv(hip_tiny, dtype):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    G, KW, seq_n, lens = 2, 31, 150, [150, 97]
    D = G * 64
    g = torch.Generator().manual_seed(17)
    x = torch.randn(len(lens) * seq_n, D, generator=g).to(dtype)
    w = (torch.randn(D, 64, KW, generator=g) / math.sqrt(64 * KW)).to(dtype)
    b = torch.randn(D, generator=g) * 0.1
    res = torch.randn(len(lens) * seq_n, D, generator=g).to(dtype)
    if dtype == torch.bfloat16:
        wp = w.reshape(G, 64, 64, KW).permute(0, 3, 1, 2).contiguous()
    else:
        wp = w.reshape(G, 64, 64, KW).permute(0, 3, 2, 1).contiguous()
    out = torch.zeros(len(lens) * seq_n, D, dtype=torch.float32, device=gu.DEV)
    dx, dw, db, dr = x.to(gu.DEV), wp.to(gu.DEV), b.to(gu.DEV), res.to(gu.DEV)
    sl = torch.tensor(lens, dtype=torch.int32, device=gu.DEV)
    a = rt.vv_posconv_args()
    a.dtype = rt.VV_BF16 if dtype == torch.bfloat16 else rt.VV_F32
    a.out_dtype = rt.VV_F32
    a.in_, a.ld_in, a.W, a.bias, a.out, a.ld_out, a.resid, a.ld_resid = dx.data_ptr(), D, dw.data_ptr(), db.data_ptr(), out.data_ptr(), D, dr.data_ptr(), D
    a.n_seq, a.seq_n, a.groups, a.KW, a.B, a.seq_len = len(lens), seq_n, G, KW, len(lens), sl.data_ptr()
    gu.check(eng, eng.lib.vv_posconv(eng.ctx, C.byref(a), gu.stream()))
    torch.cuda.synchronize()
    for s, L in enumerate(lens):
        xs = x.float()[s * seq_n: s * seq_n + L]
        ref = F.mish(F.conv1d(xs.t().unsqueeze(0), w.float(), b, padding=KW // 2, groups=G)).squeeze(0).t() + res.float()[s * seq_n: s * seq_n + L]
        got = out[s * seq_n: s * seq_n + L]
        assert gu.rel_err(got, ref) < (5e-3 if dtype == torch.bfloat16 else TOL_F32), gu.rel_err(got, ref)


# ------------------------------------------------------------------------------------ vocoder convs
def _pack_conv(w):        # torch [Cout][Cin][KW] -> [Cin_pad8][KW][Cout_pad64]
    cout, cin, kw = w.shape
    t = torch.zeros(((cin + 7) // 8 * 8, kw, (cout + 63) // 64 * 64))
    t[:cin, :, :cout] = w.permute(1, 2, 0)
    return t


def _split_conv_weights(eng, rt, gu, dw):
    """vv_conv_split_weights on a packed fp32 slab [Cin_pad][KW][rows_pad] (device) -> the x3 slab (device uint16 tensor)."""
    cin_pad, kw, rows_pad = dw.shape
    nbytes = int(eng.lib.vv_conv_split_bytes(cin_pad, kw, rows_pad))
    assert nbytes == (cin_pad + 15) // 16 * kw * 3 * rows_pad * 16 * 2
    wb = torch.zeros(nbytes // 2, dtype=torch.int16, device=gu.DEV)
    gu.check(eng, eng.lib.vv_conv_split_weights(eng.ctx, dw.data_ptr(), cin_pad, kw, rows_pad, wb.data_ptr(), gu.stream()))
    torch.cuda.synchronize()
    return wb


def _run_conv(eng, rt, gu, x, wp, bias, cout, T_out, KW, dil, up, resid=None, pre_slope=1.0, scale=1.0, accumulate=0, out0=None, lens=None,
              x3=False):
    B, cin, T_in = x.shape
    out = out0.clone().to(gu.DEV) if out0 is not None else torch.zeros(B, cout, T_out, device=gu.DEV)
    dx, dw, db = x.to(gu.DEV), wp.to(gu.DEV), bias.to(gu.DEV)
    dr = resid.to(gu.DEV) if resid is not None else None
    dl = torch.tensor(lens, dtype=torch.int32, device=gu.DEV) if lens is not None else None
    a = rt.vv_conv_args()
    a.in_, a.W, a.bias, a.out = dx.data_ptr(), dw.data_ptr(), db.data_ptr(), out.data_ptr()
    a.resid = dr.data_ptr() if dr is not None else None
    a.B, a.Cin, a.Cout, a.T_in, a.T_out, a.KW, a.dil = B, cin, cout, T_in, T_out, KW, dil
    a.transposed, a.up = (1 if up else 0), up
    a.rows_total = cout * up if up else cout
    a.rows_pad = (a.rows_total + 63) // 64 * 64
    a.accumulate, a.pre_slope, a.out_scale = accumulate, pre_slope, scale
    a.len_in = dl.data_ptr() if dl is not None else None
    if x3:
        wb = _split_conv_weights(eng, rt, gu, dw)
        a.W_x3 = wb.data_ptr()
        a.wg_rows = x3 if (isinstance(x3, int) and not isinstance(x3, bool)) else 0      # 128: the 8-wave workgroup form; -1: never the streaming up-sampler
    gu.check(eng, eng.lib.vv_conv1d(eng.ctx, C.byref(a), gu.stream()))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("x3", [False, True, 128])         # f32 MFMA / x3 (4-wave workgroups) / x3 with 8-wave 128-row workgroups
@pytest.mark.parametrize("KW,dil", [(3, 1), (3, 5), (7, 3), (11, 1), (11, 5), (7, 1)])
@pytest.mark.parametrize("cin,cout,T", [(20, 24, 300), (64, 64, 517), (100, 128, 40)])
def test_conv1d_mfma(hip_tiny, KW, dil, cin, cout, T, x3):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(KW * 100 + dil + cin)
    B = 2
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin, KW, generator=g) / math.sqrt(cin * KW)
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, T, generator=g)
    prev = torch.randn(B, cout, T, generator=g)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=dil, padding=dil * (KW - 1) // 2)
    got = _run_conv(eng, rt, gu, x, _pack_conv(w), b, cout, T, KW, dil, 0, pre_slope=0.1, x3=x3)
    assert gu.rel_err(got, ref) < TOL_F32
    ref2 = prev + (ref + res) / 3.0
    got2 = _run_conv(eng, rt, gu, x, _pack_conv(w), b, cout, T, KW, dil, 0, resid=res, pre_slope=0.1, scale=1.0 / 3.0, accumulate=1, out0=prev, x3=x3)
    assert gu.rel_err(got2, ref2) < TOL_F32


def test_conv_x3_split_is_exact_and_products_keep_fp32_fidelity(hip_tiny):
    """The x3 form (vv_vocoder_x3.hip): (a) vv_conv_split_weights is an error-free transformation -- the three bf16 pieces of
    every weight sum back to the fp32 weight EXACTLY, each piece has at most 8 significant bits by construction, zero padding
    stays zero; (b) a conv through the six piece products is as close to the float64 result as the v_mfma_f32_32x32x2_f32 path
    (the two differ from float64 by fp32 accumulation rounding; the dropped piece products are <= 2^-23 of a term)."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(4242)
    cin, cout, KW, T, dil = 100, 128, 7, 700, 3
    w = torch.randn(cout, cin, KW, generator=g) / math.sqrt(cin * KW)
    ws_ = w.clone()
    ws_[0, 0, 0], ws_[1, 0, 0], ws_[2, 0, 0] = 1.0 + 2.0 ** -23, -3.0e-7, 16777215.0        # low bit set / tiny / all 24 bits set
    wp = _pack_conv(ws_).to(gu.DEV)
    wb = _split_conv_weights(eng, rt, gu, wp)
    cin_pad, _, rows_pad = wp.shape
    pieces = (wb.view(torch.int16).to(torch.int32) << 16).view(torch.float32).reshape((cin_pad + 15) // 16, KW, 3, 2, rows_pad, 8)
    back = pieces.double().sum(2).permute(0, 2, 4, 1, 3).reshape(-1, KW, rows_pad)                   # [chunk][octet][8] = c (padded to 16), [kw][row]
    assert torch.equal(back[:cin_pad], wp.double()), "h + m + l must reproduce every fp32 weight exactly"
    assert not bool(back[cin_pad:].any()), "channels past Cin_pad are zero"
    mant = pieces.abs().view(torch.int32) & 0xFFFF
    assert not bool(mant.any()), "every piece is a bf16 value (low 16 bits of its fp32 pattern are zero)"
    x = torch.randn(2, cin, T, generator=g) * 3
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=dil, padding=dil * (KW - 1) // 2)
    d32 = _run_conv(eng, rt, gu, x, _pack_conv(w), b, cout, T, KW, dil, 0, pre_slope=0.1).cpu().double() - ref
    dx3 = _run_conv(eng, rt, gu, x, _pack_conv(w), b, cout, T, KW, dil, 0, pre_slope=0.1, x3=True).cpu().double() - ref
    e32, ex3, r32, rx3 = float(d32.abs().max()), float(dx3.abs().max()), float(d32.pow(2).mean().sqrt()), float(dx3.pow(2).mean().sqrt())
    print(f"\n[conv x3] vs float64 (|ref| max {float(ref.abs().max()):.2f}): f32 MFMA max {e32:.3e} rms {r32:.3e}; 3-way bf16 split max {ex3:.3e} rms {rx3:.3e}")
    assert ex3 <= 2.0 * e32 and rx3 <= 2.0 * r32


@pytest.mark.parametrize("KW,dil", [(3, 1), (3, 5), (7, 3), (11, 1), (11, 5)])
@pytest.mark.parametrize("C_,T", [(32, 1000), (64, 517), (64, 116), (32, 131)])
def test_mrf_resblock_fused_is_bit_identical_to_two_convs(hip_tiny, KW, dil, C_, T):
    """K12 (SURVEY 8a / 8b vv_mrf_resblock): conv1 -> LeakyReLU -> conv2 (+ residual, MRF scale, accumulate) through LDS in one
    launch.  Same contraction order and epilogue arithmetic as the per-conv kernel, so the result equals vv_conv1d(conv1) +
    vv_conv1d(conv2, resid = y) BIT FOR BIT (ragged lengths, tile seams and the accumulate form included), and matches torch
    within the fp32 kernel tolerance."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(KW * 1000 + dil * 10 + C_ + T)
    B = 3
    lens = [T, max(1, T // 3), max(1, T - 7)]
    y = torch.randn(B, C_, T, generator=g)
    w1 = torch.randn(C_, C_, KW, generator=g) * 1.4 / math.sqrt(C_ * KW)
    w2 = torch.randn(C_, C_, KW, generator=g) * 0.45 / math.sqrt(C_ * KW)
    b1, b2 = torch.randn(C_, generator=g) * 0.1, torch.randn(C_, generator=g) * 0.1
    prev = torch.randn(B, C_, T, generator=g)
    p1, p2 = _pack_conv(w1), _pack_conv(w2)
    for accumulate, scale in ((0, 1.0), (1, 1.0 / 3.0)):
        t1 = _run_conv(eng, rt, gu, y, p1, b1, C_, T, KW, dil, 0, pre_slope=0.1, lens=lens)
        two = _run_conv(eng, rt, gu, t1.cpu(), p2, b2, C_, T, KW, 1, 0, resid=y, pre_slope=0.1, scale=scale, accumulate=accumulate,
                        out0=prev if accumulate else None, lens=lens)
        out = (prev.clone() if accumulate else torch.zeros(B, C_, T)).to(gu.DEV)
        dev = [t.to(gu.DEV) for t in (y, p1, b1, p2, b2)]
        dl = torch.tensor(lens, dtype=torch.int32, device=gu.DEV)
        a = rt.vv_mrf_args()
        a.y, a.W1, a.b1, a.W2, a.b2, a.out = [t.data_ptr() for t in dev] + [out.data_ptr()]
        a.B, a.C, a.T, a.KW, a.dil, a.rows_pad, a.accumulate, a.slope, a.out_scale, a.len_in = B, C_, T, KW, dil, 64, accumulate, 0.1, scale, dl.data_ptr()
        gu.check(eng, eng.lib.vv_mrf_resblock(eng.ctx, C.byref(a), gu.stream()))
        torch.cuda.synchronize()
        assert torch.equal(out, two), float((out - two).abs().max())
        for i, L in enumerate(lens):
            yi = y[i:i + 1, :, :L]
            ref = F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(yi, 0.1), w1, b1, dilation=dil, padding=dil * (KW - 1) // 2), 0.1), w2, b2,
                           padding=(KW - 1) // 2) + yi
            ref = ref * scale + (prev[i:i + 1, :, :L] if accumulate else 0)
            assert gu.rel_err(out[i:i + 1, :, :L], ref) < TOL_F32
    a.C = 128
    assert eng.lib.vv_mrf_resblock(eng.ctx, C.byref(a), gu.stream()) != 0            # wider stages are refused (intermediate does not fit)


@pytest.mark.parametrize("x3", [False, True])
def test_conv1d_length_mask(hip_tiny, x3):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 300, generator=g)
    w = torch.randn(16, 16, 7, generator=g) / 10
    b = torch.zeros(16)
    lens = [300, 123]
    got = _run_conv(eng, rt, gu, x, _pack_conv(w), b, 16, 300, 7, 3, 0, lens=lens, x3=x3)
    for i, L in enumerate(lens):
        ref = F.conv1d(x[i:i + 1, :, :L], w, b, dilation=3, padding=9)
        assert gu.rel_err(got[i:i + 1, :, :L], ref) < TOL_F32


@pytest.mark.parametrize("x3", [False, True, 128])
@pytest.mark.parametrize("u,cin,cout,T", [(8, 32, 16, 70), (2, 16, 8, 300), (8, 64, 32, 257), (2, 128, 64, 40)])
def test_conv_transpose_polyphase(hip_tiny, u, cin, cout, T, x3):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(u * 31 + cin)
    B = 2
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cin, cout, 2 * u, generator=g) / math.sqrt(2 * cin)
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=u // 2)
    rows = cout * u
    wp = torch.zeros(((cin + 7) // 8 * 8, 2, (rows + 63) // 64 * 64))
    wp[:cin, :, :rows] = w.reshape(cin, cout, 2, u).permute(0, 2, 1, 3).reshape(cin, 2, rows)
    got = _run_conv(eng, rt, gu, x, wp, b, cout, T * u, 2, 1, u, pre_slope=0.1, x3=x3)
    assert ref.shape == got.shape
    assert gu.rel_err(got, ref) < TOL_F32


@pytest.mark.parametrize("cin,cout,T,lens", [(64, 32, 1, None), (64, 32, 63, None), (64, 32, 64, None), (64, 32, 65, None), (64, 32, 1000, [1000, 333, 1]),
                                             (128, 64, 40, None), (128, 64, 777, [777, 64, 500]), (128, 64, 4099, None)])
def test_up2_stream_equals_generic_x3(hip_tiny, cin, cout, T, lens):
    """Round 5: the x2 up-samplers (stages 2 and 3 of the vocoder: 128 -> 64 and 64 -> 32 channels) run a streaming kernel of their own
    (up2_stream_x3_kernel: registers instead of an LDS window, waves as independent workers, paired 8-byte stores).  It keeps the generic
    x3 kernel's contraction order, so it must be BIT-IDENTICAL to it (vv_conv_args.wg_rows = -1 forces the generic one), at every edge:
    T = 1, column blocks of exactly / just over 64, ragged valid lengths -- and right against torch."""
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(cin + T)
    B = 3 if lens else 2
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cin, cout, 4, generator=g) / math.sqrt(2 * cin)
    b = torch.randn(cout, generator=g) * 0.1
    rows = cout * 2
    wp = torch.zeros((cin, 2, (rows + 63) // 64 * 64))
    wp[:, :, :rows] = w.reshape(cin, cout, 2, 2).permute(0, 2, 1, 3).reshape(cin, 2, rows)
    stream = _run_conv(eng, rt, gu, x, wp, b, cout, T * 2, 2, 1, 2, pre_slope=0.1, lens=lens, x3=True)
    generic = _run_conv(eng, rt, gu, x, wp, b, cout, T * 2, 2, 1, 2, pre_slope=0.1, lens=lens, x3=-1)
    assert torch.equal(stream, generic)
    for i in range(B):
        L = lens[i] if lens else T
        xi = x[i:i + 1].clone()
        xi[:, :, L:] = 0
        ref = F.conv_transpose1d(F.leaky_relu(xi, 0.1), w, b, stride=2, padding=1)
        assert gu.rel_err(stream[i:i + 1], ref) < TOL_F32
    # a destination with a guard band: nothing is written outside [B][Cout][2 T]
    out0 = torch.full((B, cout, T * 2), 7.0)
    again = _run_conv(eng, rt, gu, x, wp, b, cout, T * 2, 2, 1, 2, pre_slope=0.1, lens=lens, x3=True, out0=out0)
    assert torch.equal(again, stream)


def test_conv_post_pcm(hip_tiny):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(23)
    B, Cc, T = 2, 32, 3000
    x = torch.randn(B, Cc, T, generator=g)
    w = torch.randn(1, Cc, 7, generator=g) / math.sqrt(Cc * 7) * 0.7
    bias = 0.05
    ref = torch.tanh(F.conv1d(F.leaky_relu(x, 0.01), w, torch.tensor([bias]), padding=3)).reshape(B, T)
    pcm = torch.zeros(B, T, dtype=torch.int16, device=gu.DEV)
    wave = torch.zeros(B, T, device=gu.DEV)
    dx, dw = x.to(gu.DEV), w.reshape(Cc, 7).contiguous().to(gu.DEV)
    gu.check(eng, eng.lib.vv_conv_post(eng.ctx, dx.data_ptr(), dw.data_ptr(), bias, pcm.data_ptr(), T, wave.data_ptr(), B, Cc, T, 7, 0.01, None, gu.stream()))
    torch.cuda.synchronize()
    assert gu.rel_err(wave, ref) < TOL_F32
    ref_pcm = torch.clamp(ref * 32767.0, -32768.0, 32767.0).to(torch.int16)
    assert int((pcm.cpu().int() - ref_pcm.int()).abs().max()) <= 1      # +-1 LSB


# ------------------------------------------------------------------------------------ mel front end
def test_mel_frontend(hip_tiny, tiny_setup):
    rt, gu = _imports()
    spec, _, orc = tiny_setup
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(31)
    lens = [256 * 20, 256 * 13 + 77]
    S = max(lens)
    audio = (torch.randn(2, S, generator=g) * 4000).clamp(-30000, 30000).to(torch.int16)
    t = torch.arange(S) / 24000.0
    audio[0] = (torch.sin(2 * math.pi * 440 * t) * 12000 + torch.sin(2 * math.pi * 3000 * t) * 3000).to(torch.int16)
    F_max = S // 256 + 1
    mel = torch.zeros(2, F_max, spec.n_mel, device=gu.DEV)
    da = audio.to(gu.DEV)
    dl = torch.tensor(lens, dtype=torch.int32, device=gu.DEV)
    gu.check(eng, eng.lib.vv_mel(eng.ctx, da.data_ptr(), S, dl.data_ptr(), mel.data_ptr(), 2, F_max, gu.stream()))
    torch.cuda.synchronize()
    for i, L in enumerate(lens):
        ref = orc.mel(audio[i, :L])
        got = mel[i, : ref.shape[0]].cpu()
        assert ref.shape[0] == L // 256 + 1
        # linear-domain check (1e-4 of the frame peak) + log-domain check where the bin is not ~silent:
        # log() of a near-zero bin amplifies the fp32 DFT-vs-FFT rounding difference without bound
        lin_g, lin_r = got.exp(), ref.exp()
        assert float((lin_g - lin_r).abs().max()) < 1e-4 * float(lin_r.max()), float((lin_g - lin_r).abs().max())
        loud = lin_r > 1e-3 * lin_r.max()
        assert float((got - ref)[loud].abs().max()) < 2e-3, float((got - ref)[loud].abs().max())
        if i == 0:   # known answer: a 440 Hz tone peaks in the mel bin whose centre is nearest 440 Hz
            peak = int(ref[10].argmax())
            assert int(got[10].argmax()) == peak


def test_cfg_euler(hip_tiny):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(41)
    BN, M, ldp = 77, 100, 128
    x = torch.randn(BN, M, generator=g)
    pred = torch.randn(2 * BN, ldp, generator=g)
    ref = x + 0.03 * (pred[:BN, :M] + (pred[:BN, :M] - pred[BN:, :M]) * 2.0)
    dx, dp = x.clone().to(gu.DEV), pred.to(gu.DEV)
    gu.check(eng, eng.lib.vv_cfg_euler(eng.ctx, dx.data_ptr(), dp.data_ptr(), ldp, BN, M, 2.0, 0.03, gu.stream()))
    torch.cuda.synchronize()
    assert gu.rel_err(dx, ref) < 1e-6


@pytest.mark.parametrize("B,Cc,T,G", [(2, 64, 300, 8), (1, 32, 4097, 32), (3, 48, 50, 1)])
def test_groupnorm(hip_tiny, B, Cc, T, G):
    rt, gu = _imports()
    eng = hip_tiny["f32"]
    g = torch.Generator().manual_seed(B * 100 + G)
    x = torch.randn(B, Cc, T, generator=g) * 3 + 1.5
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    y = torch.zeros(B, Cc, T, device=gu.DEV)
    dx, dg, db = x.to(gu.DEV), gamma.to(gu.DEV), beta.to(gu.DEV)
    for act, fn in ((0, lambda v: v), (4, F.mish)):
        gu.check(eng, eng.lib.vv_groupnorm(eng.ctx, dx.data_ptr(), y.data_ptr(), dg.data_ptr(), db.data_ptr(), B, Cc, T, G, 1e-5, act, gu.stream()))
        torch.cuda.synchronize()
        ref = fn(F.group_norm(x, G, gamma, beta, eps=1e-5))
        assert gu.rel_err(y, ref) < 1e-5
    # data with a mean far from zero (the shifted one-pass variance must not cancel) against a float64 reference
    xb = (x * 0.01 + 300.0).contiguous()
    gu.check(eng, eng.lib.vv_groupnorm(eng.ctx, xb.to(gu.DEV).data_ptr(), y.data_ptr(), dg.data_ptr(), db.data_ptr(), B, Cc, T, G, 1e-5, 0, gu.stream()))
    torch.cuda.synchronize()
    ref64 = F.group_norm(xb.double(), G, gamma.double(), beta.double(), eps=1e-5)
    assert gu.rel_err(y, ref64) < 2e-3          # the INPUT's fp32 rounding at 300 +- 0.03 already costs ~1e-3 of the normalised value


# ------------------------------------------------------------------ N3: opt-in polyphase resampler (the default ingest is in test_ingest_gpu.py)
@pytest.mark.parametrize("src", [48000, 16000, 44100, 22050])
def test_resample_poly_matches_host_mirror(hip_tiny, src):
    """vv_resample_poly vs the host's polyphase resampler (core/audio_processor.py::_resample_polyphase).  Opt-in only: the
    default ingest is the reference's audioop arithmetic (tests/test_ingest_gpu.py); this parity is GPU-vs-host-mirror."""
    import torch
    from vietvoice_tts_amd.core.audio_processor import _resample_polyphase as _resample
    from vietvoice_tts_amd.voice_bank import resample_design
    eng = hip_tiny["f32"]
    rng = np.random.default_rng(src)
    x = (rng.standard_normal(src // 3 + 17) * 5000).astype(np.float32)
    want = _resample(x, src, 24000)
    taps, up, down, skip = resample_design(src, 24000)
    n_out = -(-(x.size * up) // down)
    got = eng.resample_poly(torch.from_numpy(x).cuda(), torch.from_numpy(taps).cuda(), up, down, skip, n_out).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-3 * 1 + 1e-6 * np.abs(want).max()          # f64 accumulate both sides, f32 rounding of the result

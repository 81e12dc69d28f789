#!/usr/bin/env python3
"""Per-shape vocoder conv microbenchmark through the C ABI (vv_conv1d): the f32-MFMA kernel (K11) against the 3-way bf16 split
kernel (K11x, vv_vocoder_x3.hip) at the headline decode shapes (B = 32 x 1037 frames).  CONV_AB_LIBS=libA.so,libB.so times
the x3 kernel of several builds in one process, interleaved."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vietvoice_tts_amd import runtime as rt
from vietvoice_tts_amd.model_spec import ModelSpec, make_synthetic_weights

DEV = "cuda:0"
B = int(os.environ.get("CONV_B", "32"))
F0 = 1037
# (name, Cin, Cout, T_in, KW, dil, up)
SHAPES = [("pre k7", 100, 512, F0, 7, 1, 0), ("up0 x8", 512, 256, F0, 2, 1, 8)]
for s, (c, t) in enumerate([(256, F0 * 8), (128, F0 * 64), (64, F0 * 128), (32, F0 * 256)]):
    for kw, dil in ((3, 1), (3, 5), (7, 3), (11, 1), (11, 5)):
        SHAPES.append((f"res{s} C{c} k{kw} d{dil}", c, c, t, kw, dil, 0))
    if s < 3:
        u = (8, 2, 2)[s]
        SHAPES.append((f"up{s + 1} x{u}", c, c // 2, t, 2, 1, u))
sel = os.environ.get("CONV_SHAPES")
if sel:
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in sel.split(","))]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def engines():
    libs = [p for p in os.environ.get("CONV_AB_LIBS", "").split(",") if p]
    spec = ModelSpec.tiny()
    w = make_synthetic_weights(spec)
    if not libs:
        return [("default", rt.HipSynth(spec, w, acoustic_dtype="f32", nfe_step=2))]
    out = []
    for p in libs:
        rt._lib = None
        rt._lib = rt.load_library(p)
        out.append((os.path.basename(p), rt.HipSynth(spec, w, acoustic_dtype="f32", nfe_step=2)))
    return out


def main():
    engs = engines()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    tot = {}
    for name, cin, cout, T, kw, dil, up in SHAPES:
        rows = cout * up if up else cout
        rows_pad = (rows + 63) // 64 * 64
        cin_pad = (cin + 7) // 8 * 8
        x = torch.randn(B, cin, T, generator=g).to(DEV)
        wp = (torch.randn(cin_pad, kw, rows_pad, generator=g) / math.sqrt(cin * kw)).to(DEV)
        bias = torch.zeros(cout, device=DEV)
        T_out = T * up if up else T
        out = torch.zeros(B, cout, T_out, device=DEV)
        flops = 2.0 * B * T_out * cout * cin * (2 if up else kw)
        line = f"{name:18s} T_out {T_out:7d}"
        for en, eng in engs:
            nb = int(eng.lib.vv_conv_split_bytes(cin_pad, kw, rows_pad))
            wb = torch.zeros(nb // 2, dtype=torch.int16, device=DEV)
            assert eng.lib.vv_conv_split_weights(eng.ctx, wp.data_ptr(), cin_pad, kw, rows_pad, wb.data_ptr(), st) == 0
            for mode in ((0, 1, 2, 3) if en == engs[0][0] else (1,)):      # 0 f32 MFMA, 1 x3 (auto: the x2 up-samplers stream), 2 x3 with 128-row workgroups, 3 x3 generic kernel forced
                a = rt.vv_conv_args()
                a.in_, a.W, a.bias, a.out = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr()
                a.B, a.Cin, a.Cout, a.T_in, a.T_out, a.KW, a.dil = B, cin, cout, T, T_out, kw, dil
                a.transposed, a.up, a.rows_total, a.rows_pad = (1 if up else 0), up, rows, rows_pad
                a.pre_slope, a.out_scale = 0.1, 1.0
                a.W_x3 = wb.data_ptr() if mode else None
                a.wg_rows = (0, 0, 128, -1)[mode]
                if (mode == 2 and rows <= 64) or (mode == 3 and up != 2):
                    continue
                assert eng.lib.vv_conv1d(eng.ctx, C.byref(a), st) == 0, eng.lib.vv_last_error(eng.ctx)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    eng.lib.vv_conv1d(eng.ctx, C.byref(a), st)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                key = (en, mode)
                tot[key] = tot.get(key, 0.0) + ms
                line += f" | {en if len(engs) > 1 else ''} {('f32', 'x3', 'x3/128', 'x3-generic')[mode]} {ms:6.3f} {flops / ms / 1e9:5.1f}"
        print(line, flush=True)
    print("sum of shapes (one launch each):", {f"{k[0]}/{('f32', 'x3', 'x3-128rows (>64-row shapes only)', 'x3-generic (x2 up-samplers only)')[k[1]]}": round(v, 3) for k, v in tot.items()})


main()

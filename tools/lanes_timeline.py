#!/usr/bin/env python3
"""Overlap picture of a two-lane step from a rocprofv3 rocpd database (--kernel-trace): for the middle third of the trace, per kernel
class: launches, average duration, and how the wall time divides into 0 / 1 / 2+ kernels in flight.
    lanes_timeline.py bench_results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo, hi = t0 + (t1 - t0) // 3, t0 + 2 * (t1 - t0) // 3
win = [r for r in rows if r[1] >= lo and r[2] <= hi]
ev = sorted([(r[1], 1) for r in win] + [(r[2], -1) for r in win])
depth, last, occ = 0, lo, {}
for t, d in ev:
    occ[min(depth, 2)] = occ.get(min(depth, 2), 0) + (t - last)
    depth += d; last = t
occ[0] = occ.get(0, 0) + (hi - last)
tot = sum(occ.values())
print(f"window {(hi - lo) / 1e6:.1f} ms, {len(win)} kernels" + (f", queues {sorted({r[3] for r in win})}" if qcol else ""))
for k in sorted(occ):
    print(f"  {k}{'+' if k == 2 else ''} kernels in flight: {occ[k] / 1e6:8.2f} ms ({100 * occ[k] / tot:5.1f} %)")
def cls(n):
    for key, lab in (("attn_bf16", "attention"), ("gemm_pp_kernelILi1", "gemm qkv (pp)"), ("gemm_pp_kernelILi3", "gemm gate-store (pp)"), ("gemm_pp_kernelILi0", "gemm store (pp)"),
                     ("gemm_kernel", "gemm 128"), ("ln_mod", "layernorm"), ("posconv", "posconv"), ("conv_x3", "vocoder conv")):
        if key in n:
            return lab
    return "other"
agg = {}
for r in win:
    a = agg.setdefault(cls(r[0]), [0, 0])
    a[0] += 1; a[1] += r[2] - r[1]
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:22s} {n:6d} launches, {d / n / 1e3:8.1f} us average, {d / 1e6:8.1f} ms summed")
# a sample of the timeline: 60 consecutive kernels from the middle of the window
mid = len(win) // 2
base = win[mid][1]
print("  sample (start us, duration us, queue, class):")
for r in win[mid: mid + 60]:
    print(f"    {(r[1] - base) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{r[3] if qcol else '-'}  {cls(r[0])}")

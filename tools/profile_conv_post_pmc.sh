#!/bin/bash
# HBM-side traffic of K13 (conv_post stream kernel): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over
# tools/conv_post_bench.py (program directly after `--`); FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section).
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/post_$c -- python3 $ROOT/tools/conv_post_bench.py 5 > $OUT/post_$c.log 2>&1 || { tail -5 $OUT/post_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for fn in glob.glob("$OUT/post_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(fn)):
            if "conv_post_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                v.append(float(row["Counter_Value"]))
    res[c] = (sum(v) / max(len(v), 1), len(v))
alg_r, alg_w = 32 * 32 * 265472 * 4, 32 * 265472 * 2
rd, wr = res["FETCH_SIZE"][0] * 1024 * 2, res["WRITE_SIZE"][0] * 1024
line = (f"conv_post_kernel<7, true, 8>, B = 32, C = 32, T = 265,472 ({res['FETCH_SIZE'][1]} launches): read {rd / 1e6:.1f} MB (FETCH_SIZE x 2; algorithmic {alg_r / 1e6:.1f} MB, "
        f"ratio {rd / alg_r:.3f}), write {wr / 1e6:.1f} MB (algorithmic {alg_w / 1e6:.1f} MB)")
print(line)
open("$OUT/conv_post_pmc_traffic.txt", "w").write(line + "\n")
PY
rm -rf $OUT/post_FETCH_SIZE $OUT/post_WRITE_SIZE

#!/bin/bash
# HBM-side traffic of the vocoder conv kernels per decode shape: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes
# over tools/conv_bench.py (one timed launch per shape and kernel form; program directly after `--`).
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  echo "pass $c" >> $OUT/cpmc_progress.txt
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/ctr_$c -- python3 $ROOT/tools/conv_bench.py 1 > $OUT/ctr_$c.log 2>&1 || echo "pass $c failed" >> $OUT/cpmc_progress.txt
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob("$OUT/ctr_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(fn)):
            kn = row["Kernel_Name"]
            if "conv_x3_kernel" in kn or "conv_mfma_kernel" in kn:
                form = "x3" if "conv_x3" in kn else "f32"
                if form == "x3" and "ELi8ELi" in kn.replace(" ", ""): form = "x3/128"
                key = (form, row["Grid_Size"], row["Workgroup_Size"], kn.split("<")[1].split(">")[0] if "<" in kn else kn[-40:])
                per[key][c].append(float(row["Counter_Value"]))
with open("$OUT/conv_pmc_traffic.txt", "w") as f:
    f.write("form, grid, wg, template args: FETCH_SIZE x2 (gfx950: reports half of wide coalesced reads) and WRITE_SIZE in MB per launch (KB units in the counter)\n")
    for k, v in sorted(per.items()):
        fe = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) * 1024 * 2 / 1e6
        wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) * 1024 / 1e6
        f.write(f"{k}: read {fe:9.1f} MB  write {wr:9.1f} MB  ({len(v['FETCH_SIZE'])} launches)\n")
print(open("$OUT/conv_pmc_traffic.txt").read())
PY
rm -rf $OUT/ctr_FETCH_SIZE $OUT/ctr_WRITE_SIZE

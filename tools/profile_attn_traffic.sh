#!/bin/bash
# L2-side counters of the bf16 attention kernel at the bench shape: TCC hit / miss, FETCH_SIZE, WRITE_SIZE (separate --pmc passes).
set -u
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
LIB=$ROOT/vietvoice-tts_amd/libvvtts_hip.so
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/atr_$i -- python3 $ROOT/tools/attn_ab.py 1 $LIB > $OUT/atr_$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for fn in glob.glob("$OUT/atr_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "attn_bf16_kernel" in row["Kernel_Name"]:
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {c: sum(x) / len(x) for c, x in per.items()}
with open("$OUT/attn_traffic.txt", "w") as f:
    f.write("attn_bf16_kernel, 64 sequences x 16 heads x N = 1600 (tools/attn_ab.py), rocprofv3 --pmc, mean per launch\n")
    for c in sorted(m):
        f.write(f"    {c:16s} {m[c]:.4e}  ({len(per[c])} launches)\n")
    if "TCC_HIT_sum" in m:
        f.write(f"    TCC hit rate     {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}\n")
    if "FETCH_SIZE" in m:
        f.write(f"    fabric reads     {2 * m['FETCH_SIZE'] * 1024 / 1e9:.3f} GB per launch (FETCH_SIZE KiB x 2, gfx950 correction); algorithmic: q 0.210 + k/v 0.419 GB\n")
    if "WRITE_SIZE" in m:
        f.write(f"    fabric writes    {m['WRITE_SIZE'] * 1024 / 1e9:.3f} GB per launch; algorithmic 0.210 GB\n")
print(open("$OUT/attn_traffic.txt").read())
PY
rm -rf $OUT/atr_1 $OUT/atr_2 $OUT/atr_3

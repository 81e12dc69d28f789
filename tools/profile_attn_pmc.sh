#!/bin/bash
# SQ-side counters of the bf16 attention kernel at the bench shape (tools/attn_ab.py, shipped library), one --pmc pass per group.
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  echo "pass $i: $grp" >> $OUT/apmc_progress.txt
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/apmc_$i -- python3 $ROOT/tools/attn_ab.py 1 $ROOT/vietvoice-tts_amd/libvvtts_hip.so > $OUT/apmc_$i.log 2>&1 || { echo "group $i failed" >> $OUT/apmc_progress.txt; }
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for fn in glob.glob("$OUT/apmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "attn_bf16_kernel" in row["Kernel_Name"]:
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/attn_pmc.txt", "w") as f:
    f.write("attn_bf16_kernel, 64 sequences x 16 heads x N = 1600 (tools/attn_ab.py), rocprofv3 --pmc, mean per launch\n")
    for c, xs in sorted(per.items()):
        f.write(f"    {c:28s} {sum(xs) / len(xs):.4e}  ({len(xs)} launches)\n")
print(open("$OUT/attn_pmc.txt").read())
PY
rm -rf $OUT/apmc_1 $OUT/apmc_2 $OUT/apmc_3

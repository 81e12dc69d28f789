#!/bin/bash
# SQ-side counters of the vocoder conv kernels at chosen shapes (tools/conv_bench.py, CONV_SHAPES), one --pmc pass per counter group.
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CONV_SHAPES=${CONV_SHAPES:-"res1 C128 k3 d1,res1 C128 k11 d1,res3 C32 k3 d1"}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do       # (a fifth group mixing GRBM / TCC counters hung: HBM traffic is tools/profile_conv_traffic.sh)
  i=$((i+1))
  echo "pass $i: $grp" >> $OUT/cpmc_progress.txt
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/cpmc_$i -- python3 $ROOT/tools/conv_bench.py 1 > $OUT/cpmc_$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/cpmc_$i.log; }
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/cpmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        kn = row["Kernel_Name"]
        if "conv_x3_kernel" in kn or "conv_mfma_kernel" in kn or "up2_stream" in kn:
            key = kn.split("(")[0][-60:] + " grid=" + row.get("Grid_Size", "?")
            per[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/conv_pmc.txt", "w") as f:
    for k, v in sorted(per.items()):
        f.write(k + "\n")
        for c, xs in sorted(v.items()):
            f.write(f"    {c:28s} mean {sum(xs) / len(xs):.4e} over {len(xs)} launches\n")
print(open("$OUT/conv_pmc.txt").read())
PY
rm -rf $OUT/cpmc_*

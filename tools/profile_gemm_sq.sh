#!/bin/bash
# SQ-side counters of the persistent GEMM at the block's four shapes (tools/gemm_ab.py, shipped library), one --pmc pass per group;
# GRBM_GUI_ACTIVE in its own pass gives the effective clock (sum over the 8 XCDs / 8 / kernel time).
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GEMM_AB_SHAPES=qkv_rope_rows,out_gate_store,ff1_gelu,ff2_gate_store
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/gsq_$i -- python3 $ROOT/tools/gemm_ab.py 1 $ROOT/vietvoice-tts_amd/libvvtts_hip.so > $OUT/gsq_$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/gsq_$i.log; }
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for fn in glob.glob("$OUT/gsq_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "gemm_pp_kernel" in row["Kernel_Name"]:
            mode = row["Kernel_Name"].split("gemm_pp_kernelILi")[1][:1]
            per[(mode, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for fn in glob.glob("$OUT/gsq_4/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "gemm_pp_kernel" in row["Kernel_Name"]:
            mode = row["Kernel_Name"].split("gemm_pp_kernelILi")[1][:1]
            dur[mode].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
with open("$OUT/gemm_sq_pmc.txt", "w") as f:
    f.write("gemm_pp_kernel at M = 102,400 (tools/gemm_ab.py; mode 0 = FF1 GELU store, 1 = QKV rope, 3 = out-projection and FF2 gate store), rocprofv3 --pmc, mean per launch\n")
    for k, v in sorted(per.items()):
        f.write(f"mode {k[0]} grid {k[1]}:\n")
        for c, xs in sorted(v.items()):
            f.write(f"    {c:28s} {sum(xs) / len(xs):.4e}  ({len(xs)} launches)\n")
    for m, xs in sorted(dur.items()):
        f.write(f"mode {m}: mean kernel duration under the GRBM pass {sum(xs) / len(xs) / 1e3:.1f} us ({len(xs)} launches)\n")
print(open("$OUT/gemm_sq_pmc.txt").read())
PY
rm -rf $OUT/gsq_1 $OUT/gsq_2 $OUT/gsq_3 $OUT/gsq_4

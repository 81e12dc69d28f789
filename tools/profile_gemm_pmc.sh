#!/bin/bash
# HBM-side traffic of the persistent GEMM at the four DiT-block shapes (equal launch counts = the in-bench mix of one block):
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over tools/gemm_ab.py (program directly after `--`).
# Used because the same passes over bench.py die inside librocprofiler-sdk (profiles/r02/pmc_fetch_sigsegv_stack.txt).
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GEMM_AB_SHAPES=qkv_rope_rows,out_gate_store,ff1_gelu,ff2_gate_store
LIB=$ROOT/vietvoice-tts_amd/libvvtts_hip.so
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/gpmc_fetch -- python3 $ROOT/tools/gemm_ab.py 1 $LIB > $OUT/gpmc_fetch.log 2>&1 || { tail -5 $OUT/gpmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/gpmc_write -- python3 $ROOT/tools/gemm_ab.py 1 $LIB > $OUT/gpmc_write.log 2>&1 || { tail -5 $OUT/gpmc_write.log; exit 1; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/gpmc_tcc -- python3 $ROOT/tools/gemm_ab.py 1 $LIB > $OUT/gpmc_tcc.log 2>&1 || { tail -5 $OUT/gpmc_tcc.log; exit 1; }
python3 $ROOT/tools/pmc_traffic.py $OUT/gpmc_fetch $OUT/gpmc_write $OUT/gemm_pmc_traffic.json \
    "tools/gemm_ab.py at the block's four shapes, equal launch counts (bench.py passes fault inside librocprofiler-sdk: profiles/r02/pmc_fetch_sigsegv_stack.txt)"
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$OUT/gpmc_tcc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "gemm_pp_kernel" in row["Kernel_Name"]:
            k = row["Kernel_Name"].split("gemm_pp_kernelILi")[1][:1]
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            per[k]["n_" + row["Counter_Name"]] += 1
for k, v in sorted(per.items()):
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"gemm_pp mode {k}: TCC hit {h:.3e} miss {m:.3e} hit rate {h / max(h + m, 1):.3f} over {int(v.get('n_TCC_HIT_sum', 0))} launches")
PY
rm -rf $OUT/gpmc_fetch $OUT/gpmc_write $OUT/gpmc_tcc

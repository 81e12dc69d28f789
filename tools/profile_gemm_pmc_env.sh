#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / TCC hit rate of the persistent GEMM at the block's four shapes for ONE library under the caller's environment
# (e.g. VV_GEMM_NGROUP=4 with the diagnostic build):  bash tools/profile_gemm_pmc_env.sh <tag> <lib.so>   -> gpurun_out/r04/gemm_pmc_<tag>.json
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$(cd $ROOT && realpath $2)
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GEMM_AB_SHAPES=qkv_rope_rows,out_gate_store,ff1_gelu,ff2_gate_store
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/gp_${TAG}_$c -- python3 $ROOT/tools/gemm_ab.py 1 $LIB > $OUT/gp_${TAG}_$c.log 2>&1 || { tail -5 $OUT/gp_${TAG}_$c.log; exit 1; }
done
python3 $ROOT/tools/pmc_traffic.py $OUT/gp_${TAG}_FETCH_SIZE $OUT/gp_${TAG}_WRITE_SIZE $OUT/gemm_pmc_${TAG}.json "tools/gemm_ab.py at the block's four shapes, $TAG"
rm -rf $OUT/gp_${TAG}_FETCH_SIZE $OUT/gp_${TAG}_WRITE_SIZE

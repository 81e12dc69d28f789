#!/bin/bash
# HBM-side counters over the REAL headline step (VERDICT r4 #6d), through gpurun from the repo root:  bash tools/profile_step_pmc.sh r05
# ONE attempt per counter: if the profiler faults (round 2: inside librocprofiler-sdk under bench.py), the error tail is kept and the script stops.
# Counters only (no trace domains); the program sits directly after `--`.
set -u
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/step_pmc_$C -- python3 $ROOT/tools/step_once.py --warmup 0 > $OUT/step_pmc_$C.log 2> $OUT/step_pmc_$C.err
  rc=$?; echo "step pmc $C rc=$rc"; tail -2 $OUT/step_pmc_$C.log
  if [ $rc -ne 0 ]; then tail -c 3000 $OUT/step_pmc_$C.err > $OUT/step_pmc_${C}_fault_tail.txt; echo "profiler fault: kept $OUT/step_pmc_${C}_fault_tail.txt"; exit 1; fi
done
python3 $ROOT/tools/pmc_traffic.py $OUT/step_pmc_FETCH_SIZE $OUT/step_pmc_WRITE_SIZE $OUT/gemm_pmc_traffic.json "tools/step_once.py: ONE real headline step (B = 32, 31 Euler steps + vocoder, one lane), every launch of the kernel"
rm -rf $OUT/step_pmc_FETCH_SIZE $OUT/step_pmc_WRITE_SIZE

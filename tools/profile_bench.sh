#!/bin/bash
# Round profile of the headline bench on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r02
# 1. rocprofv3 --kernel-trace --stats over bench.py  -> per-kernel summary CSV (tools/rocpd_kernel_stats.py)
# 2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, SEPARATE passes (TCC slots), counters only (no trace domains)
# The program sits directly after `--` (no env / bash -c hop: the profiler has already initialised the GPU).
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VV_BENCH_DUMP_MAPS=$OUT/bench_maps.txt
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_kt.err
rc=$?; echo "kernel-trace rc=$rc"; [ $rc -eq 0 ] || exit 1      # a failed GPU step ends the call: no further GPU step after it
DB=$(find $OUT/prof_kt -name "*.db" | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_kernel_stats.py $DB $OUT/bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rc=$?; echo "pmc fetch rc=$rc"; [ $rc -eq 0 ] || { tail -c 1500 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rc=$?; echo "pmc write rc=$rc"; [ $rc -eq 0 ] || exit 1
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/bench_pmc_traffic.json
# the per-dispatch counter CSVs are large: keep only the aggregate
rm -rf $OUT/pmc_fetch $OUT/pmc_write
tail -c 300 $OUT/pmc_fetch.err

#!/bin/bash
# Round-end evidence on the GPU box (through gpurun from the repo root):  bash tools/round_end.sh r03
# the whole -m gpu suite, the headline bench, the rocprofv3 kernel-trace summary of the same command, the other BASELINE configurations.
set -u
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -q -s > $OUT/gputest_final.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputest_final.log
tail -3 $OUT/gputest_final.log
python bench.py --steps 8 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py --workload mixed256 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_mixed256.json 2>/dev/null; echo "mixed rc=$?"
python bench.py --workload longform --steps 3 --warmup 1 > $OUT/bench_longform.json 2>/dev/null; echo "longform rc=$?"
python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_b1_bf16.json 2>/dev/null; echo "b1 bf16 rc=$?"
python bench.py --batch 1 --dtype fp32 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_b1_fp32.json 2>/dev/null; echo "b1 fp32 rc=$?"
python bench.py --workload serve > $OUT/bench_serve_c16.json 2>/dev/null; echo "serve rc=$?"
cd /tmp && export TMPDIR=/tmp
# kernel-trace summary twice: (1) every pass on ONE lane -- each kernel with the chip to itself, the durations bench.py's roofline figures
# (its one-lane HIP-event pass) must agree with; (2) the default command: warm-up and timed steps on two lanes (half-batch kernels that
# overlap), the class pass on one
export VV_BENCH_OPTIONS=lanes=1
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_kt.err
rc=$?; echo "kernel-trace (one lane) rc=$rc"; [ $rc -eq 0 ] || exit 1
unset VV_BENCH_OPTIONS
DB=$(find $OUT/prof_kt -name "*.db" | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_kernel_stats.py $DB $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof_kt
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt2 -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof_two_lanes.json 2> $OUT/prof_kt2.err
echo "kernel-trace (default: two lanes) rc=$?"
DB=$(find $OUT/prof_kt2 -name "*.db" | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_kernel_stats.py $DB $OUT/bench_kernel_stats_two_lanes.csv && python3 $ROOT/tools/lanes_timeline.py $DB > $OUT/lanes_timeline_under_rocprof.txt
rm -rf $OUT/prof_kt2

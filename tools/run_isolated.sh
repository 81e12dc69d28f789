#!/bin/bash
# Run pytest node ids one per process, stop at the first process that dies abnormally (rc >= 128: a GPU fault must not be followed by more GPU work).
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for t in "$@"; do
  echo "=== $t" >> $OUT
  timeout -k 10 300 python -m pytest "$t" -m gpu -x -q -s >> $OUT 2>&1
  rc=$?
  echo "=== rc=$rc" >> $OUT
  if [ $rc -ge 124 ]; then echo "abnormal exit $rc at $t: stopping"; exit 1; fi
done
exit 0

#!/bin/bash
# VAR=name VALUES="a b c" [ROUNDS=2] [BENCH_ARGS=...] TAG=file bash tools/knob_sweep.sh : interleaved sweep of one environment knob on the default bench line
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-knob_sweep}.txt
: > $OUT
for r in $(seq 1 ${ROUNDS:-2}); do for v in $VALUES; do
  env $VAR=$v timeout -s KILL 300 python bench.py --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v ms_per_step', d['ms_per_step'])"
done; done >> $OUT 2>&1
cat $OUT

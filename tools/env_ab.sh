#!/bin/bash
# Generic interleaved A/B of one environment switch on the default bench line: VAR=name A=0 B=1 [ROUNDS=3] [BENCH_ARGS=...] TAG=file bash tools/env_ab.sh
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-env_ab}.txt
: > $OUT
for r in $(seq 1 ${ROUNDS:-3}); do for v in $A $B; do
  env $VAR=$v timeout -s KILL 300 python bench.py --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v ms_per_step', d['ms_per_step'], 'loss', d['config'].get('final_loss'), 'kernels/step', d['config'].get('kernel_launches_per_step'))"
done; done >> $OUT 2>&1
cat $OUT

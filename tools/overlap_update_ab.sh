#!/bin/bash
# A/B of the overlapped optimiser update (CINEMA_OVERLAP_UPDATE) on the default bench line: interleaved processes, 3 rounds, 40 timed steps.
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05_q_overlap_update_ab}.txt
: > $OUT
timeout -s KILL 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "overlapped_update or recorded_step" 2>&1 | tail -1 >> $OUT
for r in 1 2 3; do for ov in 0 1; do
  CINEMA_OVERLAP_UPDATE=$ov timeout -s KILL 300 python bench.py --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap_update=$ov ms_per_step', d['ms_per_step'], 'loss', d['config'].get('final_loss'))"
done; done >> $OUT 2>&1
cat $OUT

#!/bin/bash
# A/B of the second weight-gradient stream (CINEMA_SIDE_STREAMS=1|2) on config 2 (three interleaved rounds, 40 timed steps) and config 4 (two rounds).
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05_p_side_streams_ab}.txt
: > $OUT
for r in 1 2 3; do for n in 1 2; do
  CINEMA_SIDE_STREAMS=$n timeout -s KILL 300 python bench.py --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 side_streams=$n ms_per_step', d['ms_per_step'], 'loss', d['config'].get('final_loss'))"
done; done >> $OUT 2>&1
for r in 1 2; do for n in 1 2; do
  CINEMA_SIDE_STREAMS=$n timeout -s KILL 300 python bench.py --task seg --steps 20 --warmup 5 --cpu-budget 0 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 4 side_streams=$n ms_per_step', d['ms_per_step'])"
done; done >> $OUT 2>&1
cat $OUT

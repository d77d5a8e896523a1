#!/bin/bash
# overlapped update: grid cap / stream priority sweep (one round each, 40 timed steps), baseline first and last
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05_q_overlap_update_sweep}.txt
: > $OUT
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())" >> $OUT 2>&1
run() { env "$@" timeout -s KILL 300 python bench.py --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'ms_per_step', d['ms_per_step'])"; }
for r in 1 2; do
run CINEMA_OVERLAP_UPDATE=0
for b in 128 256 512 1024 1048576; do
  run CINEMA_OVERLAP_UPDATE=1 CINEMA_OVERLAP_UPDATE_BLOCKS=$b
  run CINEMA_OVERLAP_UPDATE=1 CINEMA_OVERLAP_UPDATE_BLOCKS=$b CINEMA_UPDATE_STREAM_PRIO=low
done
done >> $OUT 2>&1
cat $OUT

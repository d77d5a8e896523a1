#!/bin/bash
mkdir -p gpurun_out
for r in 1 2 3; do for w in "0 0" "1 1"; do
  set -- $w
  CINEMA_LAX_FUSE=$1 CINEMA_LAX_HEAD=$2 timeout -s KILL 200 python bench.py --steps 80 --warmup 20 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LAX fuse/head=$1/$2 ms_per_step', d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r04_y_lax3_ab.txt

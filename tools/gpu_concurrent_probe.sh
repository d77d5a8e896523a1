#!/bin/bash
# Would two interleaved half-batches beat one batch?  Two bench processes with batch 8 side by side on the one GPU (memory-bound kernels of one can run under the
# GEMMs of the other) against one process with batch 16 / batch 8.  Dev probe; outputs gpurun_out/r04_r_concurrent.txt
mkdir -p gpurun_out
B="--steps 80 --warmup 10 --prewarm 10 --cpu-budget 0 --profile-steps 0 --no-secondary"
one() { timeout -s KILL 200 python bench.py --batch $1 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch $1: ms_per_step', d['ms_per_step'], 'samples/s', d['value'])"; }
{
one 16 single
one 8 single
one 8 concurrentA > gpurun_out/ca.txt &
one 8 concurrentB > gpurun_out/cb.txt &
wait
cat gpurun_out/ca.txt gpurun_out/cb.txt
} | tee gpurun_out/r04_r_concurrent.txt

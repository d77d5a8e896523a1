#!/bin/bash
# BK = 32 ring (CINEMA_K32_STAGES 3, default) against the double buffer (2): GEMM kernel tests under both, then the config-2 step A/B (interleaved processes, 4 rounds)
mkdir -p gpurun_out
OUT=gpurun_out/r04_au_k32_ring_step2.txt
: > $OUT
for w in 2 3; do CINEMA_K32_STAGES=$w timeout -s KILL 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" 2>&1 | tail -1 >> $OUT; done
for r in 1 2 3 4; do for w in 2 3; do
  CINEMA_K32_STAGES=$w timeout -s KILL 200 python bench.py --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 K32_STAGES=$w ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done; done >> $OUT 2>&1
cat $OUT

#!/bin/bash
mkdir -p gpurun_out
for r in 1 2; do for k in 512 768 3072; do
  CINEMA_GEMM_K32=$k timeout -s KILL 200 python bench.py --steps 40 --warmup 15 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GEMM_K32=$k ms_per_step', d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r04_k32_ab.txt

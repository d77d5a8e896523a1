#!/bin/bash
# Round-4 session B: split tail finished in the launch (reduce-scatter) - parity test, per-shape A/B, step A/B.
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 900 -k "split_tail or gemm" 2>&1 | tail -8 > gpurun_out/r04_b_tests.log
cat gpurun_out/r04_b_tests.log
python tools/tail_ab.py > gpurun_out/r04_b_tail_ab.txt 2>&1
CINEMA_TAIL_MIN_NKT=8 CINEMA_TAIL_MIN_KT=2 CINEMA_TAIL_MIN_K=512 CINEMA_GEMM_K32=0 python tools/tail_ab.py > gpurun_out/r04_b_tail_ab_k512.txt 2>&1
cat gpurun_out/r04_b_tail_ab.txt gpurun_out/r04_b_tail_ab_k512.txt
for rep in 1 2; do
for v in 0 1; do
  CINEMA_TAIL_IN_LAUNCH=$v python bench.py --steps 30 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TAIL_IN_LAUNCH=$v ms_per_step', d['ms_per_step'])"
done
done 2>&1 | tee gpurun_out/r04_b_step_ab.txt

#!/bin/bash
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_model_gpu.py tests/test_lanes_gpu.py tests/test_boundary_gpu.py tests/test_finetune_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for r in 1 2 3; do for w in 0 1; do
  CINEMA_NBR_PREFETCH=$w timeout -s KILL 200 python bench.py --steps 80 --warmup 20 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NBR_PREFETCH=$w ms_per_step', d['ms_per_step'], d['config']['final_loss'])"
done; done 2>&1 | tee gpurun_out/r04_y_nbr_ab.txt
for w in 0 1; do
  CINEMA_NBR_PREFETCH=$w timeout -s KILL 200 python bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 --steps 12 --warmup 5 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('large fp8 NBR_PREFETCH=$w ms_per_step', d['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/r04_y_nbr_ab.txt

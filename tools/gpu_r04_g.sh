#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -k "q8 or fp8" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -m gpu -q -x --timeout 900 -k "fp8_forward_path" -s 2>&1 | grep "fp8_wgrad\]\|passed\|failed\|Error" | tail -5
for w in 0 1; do
  CINEMA_FP8_WGRAD=$w timeout -s KILL 200 python bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 --steps 8 --warmup 4 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FP8_WGRAD=$w ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done 2>&1 | tee gpurun_out/r04_g_large_fp8_ab.txt
TAG=r04_g3_large_fp8_wgrad timeout -s KILL 200 bash tools/gpu_prof_large8.sh | head -40

#!/bin/bash
# long-axis stems on a stream of their own (tape.LAX_STREAM): tests that walk the whole model, then the step A/B (interleaved processes)
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_model_gpu.py tests/test_lanes_gpu.py tests/test_ddp_gpu.py tests/test_boundary_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3
for r in 1 2; do for w in 0 1; do
  CINEMA_LAX_STREAM=$w timeout -s KILL 200 python bench.py --steps 30 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LAX_STREAM=$w ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done; done 2>&1 | tee gpurun_out/r04_y_lax_ab.txt

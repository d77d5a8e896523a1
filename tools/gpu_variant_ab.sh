#!/bin/bash
# step A/B of the in-tree library against a variant build (interleaved processes, 4 rounds, 40 timed steps): VARIANT=path TAG=name [BENCH_ARGS=...] bash tools/gpu_variant_ab.sh
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r04_variant_ab}.txt
: > $OUT
timeout -s KILL 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or conv" 2>&1 | tail -1 >> $OUT
for r in 1 2 3 4; do for w in $VARIANT default; do
  timeout -s KILL 300 python tools/lib_variant_ab.py $w --steps 40 --warmup 10 --cpu-budget 0 --profile-steps 0 --no-secondary $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $w ms_per_step', d['ms_per_step'])"
done; done >> $OUT 2>&1
cat $OUT

#!/bin/bash
# GPU tests (full log kept in gpurun_out/tests_full.log, failures with context) + bench + rocprof kernel stats
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 -rf > gpurun_out/tests_full.log 2>&1
grep -v "^W2026" gpurun_out/tests_full.log | tail -${TAILN:-3}
bash tools/gpu_prof.sh 2>&1 | grep -E "${PAT:-metric|total kernel|gemm|tail|splitk}" | cut -c1-160

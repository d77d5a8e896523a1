#!/bin/bash
# Round-4 session A: the fp8-vs-oracle gradient test, 5-step and 10-step kernel stats of config 2 (do the 1070 copyBuffer calls scale with the steps?),
# config-5 kernel stats with the e4m3 data gradients on.  Outputs under gpurun_out/ (r04_*).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -q -x --timeout 900 -k "fp8" -s 2>&1 | tail -30 > gpurun_out/r04_a_fp8_tests.log
cd /tmp && export TMPDIR=/tmp
for n in 3 8; do   # + 2 warm-up steps = 5 / 10 profiled steps
  rm -rf $R/gpurun_out/prof
  CINEMA_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mae -- python $R/bench.py --steps $n --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary > $R/gpurun_out/prof_$n.log 2>&1
  (cd $R && python tools/prof_summary.py gpurun_out/prof/mae_results.db $((n + 2)) > gpurun_out/r04_a_$((n + 2))step_kernel_stats.txt)
done
rm -rf $R/gpurun_out/prof
rm -rf $R/gpurun_out/prof_fp8
CINEMA_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8 -o mae -- python $R/bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary > $R/gpurun_out/prof_fp8.log 2>&1
(cd $R && python tools/prof_summary.py gpurun_out/prof_fp8/mae_results.db 5 > gpurun_out/r04_a_large_fp8_dgrad_kernel_stats.txt)
rm -rf $R/gpurun_out/prof_fp8
cd $R
cat gpurun_out/r04_a_fp8_tests.log
grep -n "copyBuffer" gpurun_out/r04_a_5step_kernel_stats.txt gpurun_out/r04_a_10step_kernel_stats.txt
head -12 gpurun_out/r04_a_large_fp8_dgrad_kernel_stats.txt

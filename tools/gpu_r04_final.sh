#!/bin/bash
# Round-4 final evidence in one session (every step under its own timeout): full GPU test suite, smoke, default bench line, kernel stats of configs 2 / 4 / 5-fp8
# (one stream), PMC passes stamped with the library hash.  usage: GIT_HEAD=<rev> bash tools/gpu_r04_final.sh ; then tools/pull_profiles.sh r04
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --timeout 290 2>&1 | tail -3 > gpurun_out/r04_final_tests.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_final_smoke.log 2>&1
ROUND=r04 timeout -s KILL 900 bash tools/gpu_pmc_round.sh > gpurun_out/r04_pmc.log 2>&1
bash tools/pull_profiles.sh r04 > /dev/null 2>&1   # (on the box's copy: the default bench below reads the PMC files of THIS library from profiles/)
timeout -s KILL 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
CINEMA_SIDE_WGRAD=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mae -- python $R/bench.py --steps 8 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary > $R/gpurun_out/prof.log 2>&1
(cd $R && python tools/prof_summary.py gpurun_out/prof/mae_results.db 10 > gpurun_out/r04_z_kernel_stats.txt)
rm -rf $R/gpurun_out/prof $R/gpurun_out/profseg
CINEMA_SIDE_WGRAD=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profseg -o seg -- python $R/bench.py --task seg --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 > $R/gpurun_out/profseg.log 2>&1
(cd $R && python tools/prof_summary.py gpurun_out/profseg/seg_results.db 5 > gpurun_out/r04_z_seg_kernel_stats.txt)
rm -rf $R/gpurun_out/profseg
cd $R
TAG=r04_z_large_fp8 timeout -s KILL 300 bash tools/gpu_prof_large8.sh > /dev/null 2>&1
cat gpurun_out/r04_final_tests.log gpurun_out/r04_final_smoke.log | tail -5
python -c "
import json
d = json.load(open('gpurun_out/r04_bench_default.json'))
print('headline', d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'stale', d['roofline'].get('traffic_stale'))
print('config4', d['secondary']['config4']['ms_per_step'], 'config5 fp8', d['secondary']['config5_fp8']['ms_per_step'], 'bf16', d['secondary']['config5_fp8']['bf16_ms_per_step'], 'speedup', d['secondary']['config5_fp8']['fp8_speedup_over_bf16'])
"
head -8 gpurun_out/r04_z_kernel_stats.txt; tail -4 gpurun_out/r04_pmc.log

#!/bin/bash
# Round-6 final evidence in one session (every step under its own timeout): full GPU test suite, smoke, kernel stats of configs 2 / 4 / 5-fp8 (one stream), PMC passes
# stamped with the library hash - with the kernels' average durations of the kernel-stats capture merged in (roofline.hbm_kernels) - and a traffic pass over the config-5
# fp8 step (roofline.traffic of the e4m3 weight-gradient launch), then the default bench line (which reads the PMC files of THIS library from profiles/).
# usage: GIT_HEAD=<rev> bash tools/gpu_r06_final.sh ; then tools/pull_profiles.sh r06
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -s KILL 1200 python -m pytest tests -m gpu -q -x --timeout 290 2>&1 | tail -3 > gpurun_out/r06_final_tests.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_final_smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
CINEMA_SIDE_WGRAD=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mae -- python $R/bench.py --steps 8 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary > $R/gpurun_out/prof.log 2>&1
(cd $R && python tools/prof_summary.py gpurun_out/prof/mae_results.db 10 > gpurun_out/r06_z_kernel_stats.txt)
rm -rf $R/gpurun_out/prof $R/gpurun_out/profseg
CINEMA_SIDE_WGRAD=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profseg -o seg -- python $R/bench.py --task seg --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 > $R/gpurun_out/profseg.log 2>&1
(cd $R && python tools/prof_summary.py gpurun_out/profseg/seg_results.db 5 > gpurun_out/r06_z_seg_kernel_stats.txt)
rm -rf $R/gpurun_out/profseg
cd $R
TAG=r06_z_large_fp8 timeout -s KILL 300 bash tools/gpu_prof_large8.sh > /dev/null 2>&1
ROUND=r06 timeout -s KILL 900 bash tools/gpu_pmc_round.sh > gpurun_out/r06_pmc.log 2>&1
python tools/pmc_merge_durations.py gpurun_out/r06_pmc_hbm_traffic.json gpurun_out/r06_z_kernel_stats.txt >> gpurun_out/r06_pmc.log 2>&1
# config 5 (fp8): fabric traffic of the e4m3 weight-gradient launch
CINEMA_SIDE_WGRAD=0 PMC_CMD="python bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 --steps 2 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary" PMC_OUT=r06_pmc_hbm_traffic_fp8 timeout -s KILL 600 bash tools/gpu_pmc_traffic_cmd.sh >> gpurun_out/r06_pmc.log 2>&1
python - <<'PY'
import hashlib, json, os
f = "gpurun_out/r06_pmc_hbm_traffic_fp8.json"
if os.path.exists(f):
    d = json.load(open(f))
    d["so_sha256"] = hashlib.sha256(open("cinema_amd/libcinema_hip.so", "rb").read()).hexdigest()
    d["git_head"] = os.environ.get("GIT_HEAD", "unknown")
    d["command"] = "CINEMA_SIDE_WGRAD=0 python bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 --steps 2 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary"
    json.dump(d, open(f, "w"), indent=1)
PY
# stems: two-stream phase timeline of the final tree and, on the SAME box, of the stems as separate launches (CINEMA_FUSED_STEM=0 = round 5's form); one-stream stem tables; microbenchmark
BENCH_ARGS=--no-secondary PHASE_DUMP=stems timeout -s KILL 400 bash tools/gpu_timeline.sh > gpurun_out/r06_z_phase_timeline.txt 2>&1
CINEMA_FUSED_STEM=0 BENCH_ARGS=--no-secondary PHASE_DUMP=stems timeout -s KILL 400 bash tools/gpu_timeline.sh > gpurun_out/r06_z_phase_timeline_unfused_stems.txt 2>&1
TAG=r06_z BENCH_ARGS= timeout -s KILL 400 bash tools/gpu_stem_timeline.sh > /dev/null 2>&1
TAG=r06_z_unfused CINEMA_FUSED_STEM=0 timeout -s KILL 400 bash tools/gpu_stem_timeline.sh > /dev/null 2>&1
timeout -s KILL 200 python tools/bench_stem.py > gpurun_out/r06_z_bench_stem.txt 2>&1
rm -rf gpurun_out/tl gpurun_out/tl1
bash tools/pull_profiles.sh r06 > /dev/null 2>&1   # (on the box's copy: the default bench below reads the PMC files of THIS library from profiles/)
timeout -s KILL 1200 python bench.py 2>/dev/null | tail -1 > gpurun_out/r06_bench_default.json
cat gpurun_out/r06_final_tests.log gpurun_out/r06_final_smoke.log | tail -5
python -c "
import json
d = json.load(open('gpurun_out/r06_bench_default.json'))
print('headline', d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'stale', d['roofline'].get('traffic_stale'), 'fixed', d['config'].get('fixed_ms_per_step'))
print('config4', d['secondary']['config4']['ms_per_step'], 'config5 fp8', d['secondary']['config5_fp8']['ms_per_step'], 'bf16', d['secondary']['config5_fp8']['bf16_ms_per_step'], 'speedup', d['secondary']['config5_fp8']['fp8_speedup_over_bf16'])
print('hbm rows', list((d['roofline'].get('hbm_kernels') or {}).get('kernels', {}).items())[:3])
"
head -8 gpurun_out/r06_z_kernel_stats.txt; tail -6 gpurun_out/r06_pmc.log

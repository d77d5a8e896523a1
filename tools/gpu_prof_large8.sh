#!/bin/bash
# kernel stats of the config-5 shape in fp8 with the e4m3 weight gradients (one stream); TAG names the output
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04_g_large_fp8_wgrad}
rm -rf $R/gpurun_out/prof_fp8
CINEMA_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8 -o mae -- python $R/bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype fp8 --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary > $R/gpurun_out/prof_fp8.log 2>&1
(cd $R && python tools/prof_summary.py gpurun_out/prof_fp8/mae_results.db 5 > gpurun_out/${TAG}_kernel_stats.txt)
rm -rf $R/gpurun_out/prof_fp8
head -45 $R/gpurun_out/${TAG}_kernel_stats.txt

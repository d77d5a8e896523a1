#!/bin/bash
# rocprof kernel stats of the config-5 shape (ViT-Large, SAX 256x256x24 + 3 LAX 256x256, batch 8), bf16 and fp8 forward; one stream.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dt in bf16 fp8; do
  rm -rf $R/gpurun_out/prof_$dt
  CINEMA_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$dt -o mae -- python $R/bench.py --size large --sax 256,256,24 --lax 256,256 --batch 8 --dtype $dt --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary > $R/gpurun_out/prof_$dt.log 2>&1
  (cd $R && python tools/prof_summary.py gpurun_out/prof_$dt/mae_results.db 5 > gpurun_out/r03_large_${dt}_kernel_stats.txt)
  rm -rf $R/gpurun_out/prof_$dt
done
head -32 $R/gpurun_out/r03_large_bf16_kernel_stats.txt; head -32 $R/gpurun_out/r03_large_fp8_kernel_stats.txt

#!/bin/bash
# single-stream rocprofv3 kernel trace of a few replayed steps -> per-kernel totals of the stem phases (kernel durations without the other streams beside them)
# usage (on the GPU box): TAG=r06_c [CINEMA_FUSED_STEM=0] bash tools/gpu_stem_timeline.sh
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl1
CINEMA_SIDE_WGRAD=0 CINEMA_LAX_STREAM=0 rocprofv3 --kernel-trace -d $R/gpurun_out/tl1 -o mae -- python $R/bench.py --steps 6 --warmup 4 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary $BENCH_ARGS > $R/gpurun_out/tl1.log 2>&1
cd $R
tail -1 gpurun_out/tl1.log | cut -c1-120
PHASE_DUMP=stems python tools/phase_timeline.py gpurun_out/tl1/mae_results.db > gpurun_out/${TAG}_stem_single_stream.txt 2>&1
rm -rf gpurun_out/tl1

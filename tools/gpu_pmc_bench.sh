#!/bin/bash
# HBM-traffic PMC passes over the bench workload (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass and
# counter collection is never combined with the sys/runtime trace domains).  Output: gpurun_out/pmc_traffic.json
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcf $R/gpurun_out/pmcw
export CINEMA_SIDE_WGRAD=0
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --profile-steps 0 --prewarm 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcf -o t -- $CMD > $R/gpurun_out/pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcw -o t -- $CMD > $R/gpurun_out/pmcw.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmcf/t_results.db gpurun_out/pmcw/t_results.db > gpurun_out/pmc_traffic.json
head -c 3000 gpurun_out/pmc_traffic.json

#!/bin/bash
# PMC passes over the config-4 (ConvUNetR fine-tuning) step: MFMA-busy and HBM traffic per kernel (counter collection only, --kernel-trace).
# Output: gpurun_out/r02_seg_mfma_util.json, gpurun_out/r02_seg_pmc_traffic.json
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcs1 $R/gpurun_out/pmcs2 $R/gpurun_out/pmcs3
export CINEMA_SIDE_WGRAD=0
CMD="python $R/bench.py --task seg --steps 2 --warmup 1 --cpu-budget 0 --profile-steps 0 --prewarm 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmcs1 -o t -- $CMD > $R/gpurun_out/pmcs1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcs2 -o t -- $CMD > $R/gpurun_out/pmcs2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcs3 -o t -- $CMD > $R/gpurun_out/pmcs3.log 2>&1
cd $R
python tools/pmc_mfma.py $(ls gpurun_out/pmcs1/*results.db) > gpurun_out/r02_seg_mfma_util.json
python tools/pmc_traffic.py gpurun_out/pmcs2/t_results.db gpurun_out/pmcs3/t_results.db > gpurun_out/r02_seg_pmc_traffic.json
rm -rf gpurun_out/pmcs1 gpurun_out/pmcs2 gpurun_out/pmcs3
python -c "
import json
d=json.load(open('gpurun_out/r02_seg_mfma_util.json'))['kernels']
for k,v in list(d.items())[:14]: print(k[:56].ljust(56), v['launches'], v.get('duration_us_profiled'), v.get('mfma_util'))
"

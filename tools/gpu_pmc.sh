#!/bin/bash
# PMC passes for one GEMM shape; results summarised to gpurun_out/pmc_*.txt
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${MODE:-fwd}; ONLY=${ONLY:-enc fc1}
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc1 -o g -- python $R/tools/bench_gemm.py $MODE --only "$ONLY" > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc2 -o g -- python $R/tools/bench_gemm.py $MODE --only "$ONLY" > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc3 -o g -- python $R/tools/bench_gemm.py $MODE --only "$ONLY" > $R/gpurun_out/pmc3.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc1/g_results.db gpurun_out/pmc2/g_results.db gpurun_out/pmc3/g_results.db

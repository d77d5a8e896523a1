#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate counter-only passes) per kernel of any command.
# usage: PMC_CMD="python tools/x.py" PMC_OUT=name bash tools/gpu_pmc_traffic_cmd.sh   ->  gpurun_out/<name>.json
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcf $R/gpurun_out/pmcw
(cd $R && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcf -o t -- $PMC_CMD > $R/gpurun_out/pmcf.log 2>&1)
(cd $R && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcw -o t -- $PMC_CMD > $R/gpurun_out/pmcw.log 2>&1)
cd $R
python tools/pmc_traffic.py $(ls gpurun_out/pmcf/*results.db | head -1) $(ls gpurun_out/pmcw/*results.db | head -1) > gpurun_out/${PMC_OUT:-traffic_cmd}.json
grep -h "algorithmic" gpurun_out/pmcf.log
python - gpurun_out/${PMC_OUT:-traffic_cmd}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in list(d["kernels"].items())[:4]:
    print(k[:60], {a: b for a, b in v.items()})
PY
rm -rf gpurun_out/pmcf gpurun_out/pmcw

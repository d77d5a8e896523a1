#!/bin/bash
# MFMA-utilisation PMC passes over the bench workload (counter collection only: --kernel-trace, never the sys/runtime trace domains).
# Output: gpurun_out/r02_mfma_util.json (+ the counter list the box offers, for the record)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcm1 $R/gpurun_out/pmcm2
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CY[A-Z_]*\|GRBM_GUI_ACTIVE\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY\|SQ_INSTS_VALU\b" | sort -u > $R/gpurun_out/pmc_counters_available.txt
export CINEMA_SIDE_WGRAD=0
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --profile-steps 0 --prewarm 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmcm1 -o t -- $CMD > $R/gpurun_out/pmcm1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/pmcm2 -o t -- $CMD > $R/gpurun_out/pmcm2.log 2>&1
cd $R
python tools/pmc_mfma.py $(ls gpurun_out/pmcm1/*results.db gpurun_out/pmcm2/*results.db 2>/dev/null) > gpurun_out/r02_mfma_util.json
python - <<'PY' > gpurun_out/pmc_db_schema.txt 2>&1
import sqlite3, glob
for db in glob.glob("gpurun_out/pmcm1/*results.db"):
    con = sqlite3.connect(db)
    for (n, t) in con.execute("select name, type from sqlite_master where type in ('table','view')"):
        if n.startswith("rocpd_") and not n.startswith("rocpd_info"): continue
        print(t, n, [r[1] for r in con.execute(f"pragma table_info({n})")])
    print(con.execute("select * from counters_collection limit 12").fetchall())
PY
rm -rf gpurun_out/pmcm1 gpurun_out/pmcm2
python -c "
import json
d=json.load(open('gpurun_out/r02_mfma_util.json'))['kernels']
for k,v in d.items(): print(k[:56].ljust(56), v['launches'], v.get('gpu_cycles'), v.get('duration_us_profiled'), v.get('implied_clock_ghz'), v.get('mfma_util'))
"
tail -n 3 gpurun_out/pmcm1.log gpurun_out/pmcm2.log
cat gpurun_out/pmc_counters_available.txt

#!/bin/bash
# Counter passes of one round (ROUND=r04 ...) over the bench workload (one stream), all stamped with the sha256 of the library they ran:
#   gpurun_out/${ROUND}_pmc_hbm_traffic.json  (FETCH_SIZE / WRITE_SIZE, separate passes)
#   gpurun_out/${ROUND}_mfma_util.json        (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes)
# Counter collection only (--kernel-trace, never the sys / runtime trace domains).  usage: ROUND=r04 GIT_HEAD=<rev> bash tools/gpu_pmc_round.sh ; then, in the build container, tools/pull_profiles.sh copies gpurun_out/$ROUND_* into profiles/ (only gpurun_out/ travels back from the GPU box)
ROUND=${ROUND:-r04}; export ROUND
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcf $R/gpurun_out/pmcw $R/gpurun_out/pmcm1 $R/gpurun_out/pmcm2
export CINEMA_SIDE_WGRAD=0
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcf -o t -- $CMD > $R/gpurun_out/pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcw -o t -- $CMD > $R/gpurun_out/pmcw.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmcm1 -o t -- $CMD > $R/gpurun_out/pmcm1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/pmcm2 -o t -- $CMD > $R/gpurun_out/pmcm2.log 2>&1
cd $R
python tools/pmc_traffic.py $(ls gpurun_out/pmcf/*results.db | head -1) $(ls gpurun_out/pmcw/*results.db | head -1) > gpurun_out/${ROUND}_pmc_hbm_traffic.json
python tools/pmc_mfma.py $(ls gpurun_out/pmcm1/*results.db gpurun_out/pmcm2/*results.db 2>/dev/null) > gpurun_out/${ROUND}_mfma_util.json
python - <<'PY'
import hashlib, json, os
sha = hashlib.sha256(open("cinema_amd/libcinema_hip.so", "rb").read()).hexdigest()
R = os.environ["ROUND"]
for f in (f"gpurun_out/{R}_pmc_hbm_traffic.json", f"gpurun_out/{R}_mfma_util.json"):
    d = json.load(open(f))
    d["so_sha256"], d["git_head"] = sha, os.environ.get("GIT_HEAD", "unknown")
    d["command"] = "CINEMA_SIDE_WGRAD=0 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary"
    json.dump(d, open(f, "w"), indent=1)
    top = list(d["kernels"].items())[:6]
    print(f, sha[:12], [(k[:48], v.get("hbm_bytes_per_launch") or v.get("mfma_util")) for k, v in top])
PY
rm -rf gpurun_out/pmcf gpurun_out/pmcw gpurun_out/pmcm1 gpurun_out/pmcm2

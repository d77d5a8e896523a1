#!/bin/bash
# LDS / issue counters of the kernels of any command (counter collection only).  usage: PMC_CMD="python tools/x.py" PMC_FILTER=attn PMC_OUT=name bash tools/gpu_pmc_kernel.sh
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_OUT:-r03_kernel_pmc}.txt
: > $OUT
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  rm -rf $R/gpurun_out/pmcp
  (cd $R && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcp -o t -- $PMC_CMD > $R/gpurun_out/pmcp.log 2>&1)
  python - "$R/gpurun_out/pmcp" "${PMC_FILTER:-gemm}" >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    if sys.argv[2] not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    print(k, "dispatches", len(nd[k]))
    for name, v in sorted(c.items()): print(f"    {name:28s} {v / len(nd[k]):16.0f}")
PY
done
rm -rf $R/gpurun_out/pmcp
cat $OUT

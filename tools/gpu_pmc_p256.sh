#!/bin/bash
# LDS / issue counters of the persistent 256x256 GEMM per operand layout (counter collection only).  Output: gpurun_out/r03_p256_pmc.txt
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_p256_pmc.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*" | sort -u > $R/gpurun_out/pmc_counters_sq.txt
: > $OUT
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  rm -rf $R/gpurun_out/pmcp
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcp -o t -- python $R/tools/p256_probe.py $PROBE_ARGS > $R/gpurun_out/pmcp.log 2>&1
  python - "$R/gpurun_out/pmcp" >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv", glob.glob(sys.argv[1] + "/**/*", recursive=True)[:20]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    if "gemm" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    print(k, "dispatches", len(nd[k]))
    for name, v in sorted(c.items()): print(f"    {name:28s} {v / len(nd[k]):16.0f}")
PY
  tail -2 $R/gpurun_out/pmcp.log >> $OUT
done
rm -rf $R/gpurun_out/pmcp
cat $OUT

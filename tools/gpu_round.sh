#!/bin/bash
# One GPU-box session: tests, smoke, bench, rocprof kernel stats. Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -25 > gpurun_out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --steps ${STEPS:-5} --warmup 2 > gpurun_out/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o mae -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head
tail -3 gpurun_out/tests.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log

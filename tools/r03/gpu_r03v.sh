#!/bin/bash
# full GPU suite + smoke + the two fuzz tools on the final tree
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; rc=$?
tail -6 gpurun_out/r03/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
timeout -k 10 200 python tools/fuzz_trigger_pass.py 100 3 > gpurun_out/r03/fuzz_trigger.log 2>&1; rc=$?; tail -2 gpurun_out/r03/fuzz_trigger.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/fuzz_events.py 150 3 > gpurun_out/r03/fuzz_events.log 2>&1; rc=$?; tail -2 gpurun_out/r03/fuzz_events.log
exit $rc

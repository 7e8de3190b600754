#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; rc=$?
tail -6 gpurun_out/r03/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles_r03.sh

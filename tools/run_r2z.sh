export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_events.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for sp in 1 2 0; do for st in 0 1; do
echo "K2 1680 split=$sp store=$st: $(ABUB_K2_SPLIT=$sp timeout -k 5 120 ./tools/k2_microbench 3000 6 $st 1680 1050 | tail -1 | cut -c1-175)"
done; done
for sp in 1 0; do for st in 0 1; do
echo "K2 1280 split=$sp store=$st: $(ABUB_K2_SPLIT=$sp timeout -k 5 120 ./tools/k2_microbench 4000 6 $st 1280 1024 | tail -1 | cut -c1-175)"
done; done
B="python bench.py --micro-frames 0 --no-cpu-baseline --ingest-events 0 --stream-steps 0 --latency-steps 0 --min-seconds 1"
P='import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{\"metric")][-1]); print("value",round(r["value"]),"ms",round(r["ms_per_step"],3),"roof_ms",round(r["roofline"]["ms_per_launch"],3))'
for sp in 1 2 1 2; do
echo "bench 1680 K2_SPLIT=$sp: $(ABUB_K2_SPLIT=$sp timeout -k 10 300 $B --width 1680 --height 1050 2>/dev/null | python3 -c "$P")"
done
echo "bench 1280: $(timeout -k 10 300 $B 2>/dev/null | python3 -c "$P")"

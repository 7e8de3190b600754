export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
M=./tools/k2_microbench
( echo "# trigger c2 / c3"; $M 2000 5 0; ABUB_K2_CHAIN=3 $M 2000 5 0;
  echo "# store c2 / c3"; $M 2000 5 1; ABUB_K2_CHAIN=3 $M 2000 5 1;
  echo "# cycle8 trigger c2/c3, store c2"; $M 2000 5 0 1280 1024 0 1 1 8; ABUB_K2_CHAIN=3 $M 2000 5 0 1280 1024 0 1 1 8; $M 2000 5 1 1280 1024 0 1 1 8;
  echo "# nodisc trigger c2 / c3 ; store c2"; $M 2000 5 0 1280 1024 0 1 1 0 0; ABUB_K2_CHAIN=3 $M 2000 5 0 1280 1024 0 1 1 0 0; $M 2000 5 1 1280 1024 0 1 1 0 0;
  echo "# 1680 trigger, store; nodisc trigger, store"; $M 2000 5 0 1680 1050; $M 2000 5 1 1680 1050; $M 2000 5 0 1680 1050 0 1 1 0 0; $M 2000 5 1 1680 1050 0 1 1 0 0 ) > $O/micro.jsonl 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r2h/micro.jsonl'):
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    r=json.loads(l)
    print(f"   W={r['W']} store={r['store']} sig={r['sigma']} ms={r['ms_avg']:.4f} (min {r['ms_min']:.4f}) us/job={1e3*r['ms_avg']/r['frames']:.4f} frac={r['frac_of_8TBps']:.3f}")
PY

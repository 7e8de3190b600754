export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2g; mkdir -p $O
M=./tools/k2_microbench
( echo "# trigger pf1 / pf2 / c3pf1 / c3pf2"; $M 2000 5 0; ABUB_K2_CHAIN_PF=2 $M 2000 5 0; ABUB_K2_CHAIN=3 $M 2000 5 0; ABUB_K2_CHAIN=3 ABUB_K2_CHAIN_PF=2 $M 2000 5 0;
  echo "# store pf1 / pf2"; $M 2000 5 1; ABUB_K2_CHAIN_PF=2 $M 2000 5 1;
  echo "# nodisc trigger pf1 / pf2 / c3 ; store pf1/pf2"; $M 2000 5 0 1280 1024 0 1 1 0 0; ABUB_K2_CHAIN_PF=2 $M 2000 5 0 1280 1024 0 1 1 0 0; ABUB_K2_CHAIN=3 $M 2000 5 0 1280 1024 0 1 1 0 0; $M 2000 5 1 1280 1024 0 1 1 0 0; ABUB_K2_CHAIN_PF=2 $M 2000 5 1 1280 1024 0 1 1 0 0;
  echo "# nodisc sigma2 trigger (no suspects at all)"; $M 2000 5 0 1280 1024 0 2 1 0 0; ABUB_K2_CHAIN_PF=2 $M 2000 5 0 1280 1024 0 2 1 0 0;
  echo "# 1680 trigger pf1/pf2, store pf1/pf2"; $M 2000 5 0 1680 1050; ABUB_K2_CHAIN_PF=2 $M 2000 5 0 1680 1050; $M 2000 5 1 1680 1050; ABUB_K2_CHAIN_PF=2 $M 2000 5 1 1680 1050 ) > $O/micro.jsonl 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r2g/micro.jsonl'):
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    r=json.loads(l)
    print(f"   W={r['W']} store={r['store']} sig={r['sigma']} ms={r['ms_avg']:.4f} (min {r['ms_min']:.4f}) us/job={1e3*r['ms_avg']/r['frames']:.4f} frac={r['frac_of_8TBps']:.3f}")
PY

set -o pipefail
export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2d; mkdir -p $O
M=./tools/k2_microbench
( echo "# cycle8 trigger/store"; $M 2000 5 0 1280 1024 0 1 1 8; $M 2000 5 1 1280 1024 0 1 1 8;
  echo "# cycle8 rowmachine trigger"; ABUB_K2_BOUND=0 $M 2000 5 0 1280 1024 0 1 1 8;
  echo "# cycle8 sigma2 trigger"; $M 2000 5 0 1280 1024 0 2 1 8;
  echo "# normal"; $M 2000 5 0; $M 2000 5 1 ) > $O/micro.jsonl 2>&1
cat $O/micro.jsonl
bash tools/prof_k2.sh r2d_hist 2000 0 > $O/prof_hist.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r2d_hist $O/k2_hist_pmc_summary.json
bash tools/prof_k2.sh r2d_store 2000 1 > $O/prof_store.log 2>&1; python3 tools/summarize_pmc.py gpurun_out/r2d_store $O/k2_store_pmc_summary.json

M=./tools/k2_microbench
for rep in 1 2; do
for v in base sub16 sub8; do
  if [ $v = base ]; then export LD_LIBRARY_PATH=$PWD/autobub3hs_amd; elif [ $v = sub16 ]; then export LD_LIBRARY_PATH=$PWD/tools/ab; else export LD_LIBRARY_PATH=$PWD/tools/ab8; fi
  echo "# $v trigger / store (discs), k3 sigma2 discs"
  ( $M 2000 5 0; $M 2000 5 1; $M 2000 5 0 1280 1024 0 2 1 0 1 1 ) 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('   ',{k:r[k] for k in ('store','k3','ms_avg','ms_min') if k in r})"
done; done

export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
O=gpurun_out/r2q; mkdir -p $O
# two ranks sharing the one GPU of this box (gloo for the barrier / reductions): rehearsal of the self-launching multi-rank path
ABUB_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --events 30 --steps 5 --warmup 2 --micro-frames 0 --min-seconds 1 > $O/bench_g2.json 2> $O/bench_g2.err; echo "rc=$?"; tail -2 $O/bench_g2.err | cut -c1-300
python3 -c "
import json
r=json.loads(open('$O/bench_g2.json').read().strip().splitlines()[-1])
print(r['n_gpus'], r['value'], r['ms_per_step'], r['config']['timing'])
"
# nccl on one GPU with 2 ranks is not possible (one device per rank); the driver does that on an 8-GPU node

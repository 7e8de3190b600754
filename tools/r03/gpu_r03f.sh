#!/bin/bash
# power / clock under the trigger pass: rocm-smi samples while the pass loops
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
rocm-smi --showpower --showclocks --showmaxpower > $O/smi_idle.txt 2>&1
cat > /tmp/loop_pass.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from autobub3hs_amd import hip, synth
W, H, F, E, C = 1280, 1024, 41, 100, 2
dev = "cuda:0"; S = E * C
slab = torch.empty((S, F, H, W), dtype=torch.uint8, device=dev)
bgs = [synth.background(W, H, synth.BASE_SEED + c, "torch", dev) for c in range(C)]
for e in range(E):
    for c in range(C):
        spec = synth.random_spec(W, H, F, e, c, p_second=0.2)
        synth.render_event(W, H, spec, e, c, xp="torch", device=dev, out=slab[e * C + c], bg=bgs[c])
sg = torch.ones((C, H, W), dtype=torch.uint8, device=dev)
s6 = hip.sigma6(sg)
njobs = S * (F - 1)
jobs = hip.stack_jobs(S, F, 1, F - 1, 2, C, dev)
hist = torch.empty((njobs, 256), dtype=torch.int32, device=dev)
for name, opts in (("sad_k4", {"sad": 1, "chain": 4}), ("u16_k3", {"sad": 0, "chain": 3})):
    for k, v in opts.items():
        hip.k2_set_option(k, v)
    print("START", name, time.time(), flush=True)
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.diff_hist(slab, s6, jobs, W, H, store=False, hist=hist, chain=(F - 1, 2))
        e1.record(); torch.cuda.synchronize(); n += 1
        print(name, round(e0.elapsed_time(e1) / 20, 4), flush=True)
    time.sleep(1.0)
PY
( for i in $(seq 1 40); do echo "T $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk"; sleep 0.3; done ) > $O/smi_samples.txt &
SMI=$!
timeout -k 10 200 python3 /tmp/loop_pass.py > $O/loop.txt 2>&1
wait $SMI
tail -5 $O/loop.txt
grep -c T $O/smi_samples.txt

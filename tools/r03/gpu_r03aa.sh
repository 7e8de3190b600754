#!/bin/bash
# end-to-end ingest with GPU decode on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
B="python bench.py --steps 2 --warmup 1 --micro-frames 0 --no-cpu-baseline --stream-steps 0 --latency-steps 0 --regime-steps 0 --min-seconds 0 --ingest-events 96"
for g in 1 0; do
  ABUB_GPU_DECODE=$g timeout -k 10 500 $B > gpurun_out/r03/ingest_gpu$g.json 2> gpurun_out/r03/ingest_gpu$g.err; echo "gpu=$g rc=$?"
  python3 - gpurun_out/r03/ingest_gpu$g.json <<'P'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric')][-1])
i=b["config"]["ingest_inclusive"] if "ingest_inclusive" in b["config"] else b.get("ingest_inclusive")
print(json.dumps({k:i[k] for k in ("frames_per_s","frames_decoded_on_gpu","frames_decoded_on_host","seconds","batches","decode_threads")}))
P
done

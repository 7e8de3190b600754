#!/usr/bin/env python3
"""Ingestion-inclusive rate of the batched path (zip archive of PNGs -> ZipParser -> files uploaded and decoded on the GPU, or
with ABUB_GPU_DECODE=0 decode threads -> pinned batches -> GPU pipeline) for several reading / decode-thread counts; the archive
is written once.
usage: python3 tools/ingest_bench.py [--events 96] [--width 1280 --height 1024] threads [threads ...]"""
import argparse, io, json, os, shutil, sys, tempfile, time, zipfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
from autobub3hs_amd import host, synth

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=96)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--height", type=int, default=1024)
ap.add_argument("--frames", type=int, default=41)
ap.add_argument("--host-threads", type=int, default=16)
ap.add_argument("threads", nargs="+", type=int)
a = ap.parse_args()
W, H, F, E, C = a.width, a.height, a.frames, a.events, 2
run_id = "20200925_0"
tmp = tempfile.mkdtemp(prefix="abub_ingest_")
try:
    t0 = time.perf_counter()

    def enc(job):
        e, c = job
        st = synth.render_event(W, H, synth.random_spec(W, H, F, e, c, p_second=0.2), e, c)
        out = []
        for k in range(F):
            b = io.BytesIO()
            Image.fromarray(st[k]).save(b, format="PNG", compress_level=1)
            out.append((e, c, k, b.getvalue()))
        return out

    with ThreadPoolExecutor(min(64, len(os.sched_getaffinity(0)))) as ex:
        blobs = [x for l in ex.map(enc, [(e, c) for e in range(E) for c in range(C)]) for x in l]
    zpath = os.path.join(tmp, run_id + ".zip")
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_STORED) as z:
        for e in range(E):
            z.writestr(f"{run_id}/{e}/", b"")
            z.writestr(f"{run_id}/{e}/Images/", b"")
        for e, c, k, data in blobs:
            z.writestr(f"{run_id}/{e}/Images/cam{c}_image{30 + k}.png", data)
    del blobs
    print(json.dumps({"archive_s": round(time.perf_counter() - t0, 1), "MB": round(os.path.getsize(zpath) / 1e6)}), flush=True)
    run = host.Run(kind="zip", run_folder=os.path.join(tmp, run_id))
    t1 = time.perf_counter()
    tr = [run.train(c, shape=(H, W)) for c in range(C)]
    t_train = time.perf_counter() - t1
    assert all(t[0] == 0 for t in tr)
    for rep in range(2):
        for nt in a.threads:
            t2 = time.perf_counter()
            st = run.run_batched(C, tmp + "/", run_id, 30, nthreads=a.host_threads, decode_threads=nt)
            dt = time.perf_counter() - t2
            print(json.dumps({"decode_threads": nt, "frames_per_s": round(E * C * F / dt), "detect_total_s": round(dt, 3),
                              "decode_s": round(st["decode_s"], 3), "gpu_s": round(st["gpu_s"], 3), "list_s": round(st["list_s"], 3),
                              "batches": int(st["batches"]), "train_s": round(t_train, 2), "frames_decoded_on_gpu": int(st["frames_gpu_decoded"]),
                              "gpu_decode_s": round(st["gpudecode_s"], 3)}), flush=True)
    run.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)

#!/usr/bin/env python3
"""Times abub_png_decode_dev on a batch of full-size frames (GPU box).  Usage: python tools/png_bench.py [nframes] [level] [W H]"""
import io
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from autobub3hs_amd import hip, synth, _lib  # noqa: E402
from PIL import Image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 656
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
F = 41
spec = synth.random_spec(W, H, F, 3, 0)
fr = np.asarray(synth.render_event(W, H, spec, 3, 0))
enc = []
for k in range(F):
    b = io.BytesIO()
    Image.fromarray(fr[k]).save(b, format="PNG", compress_level=level)
    enc.append(b.getvalue())
files = [enc[k % F] for k in range(n)]
print("encoded", sum(len(f) for f in files) / n / 1e6, "MB per frame", flush=True)
# one call through the Python helper for the check, then timed calls on resident buffers
if not os.environ.get("ABUB_PNG_DEBUG"):
    out, st = hip.png_decode(files, W, H)
    torch.cuda.synchronize()
    assert (st == 0).all(), st
    for k in range(0, n, max(1, n // 7)):
        assert np.array_equal(out[k].cpu().numpy(), fr[k % F]), k
    del out
# resident buffers
frames_np = np.zeros((n, 8), dtype=np.uint32)
segs, blob, zoff = [], bytearray(), 0
P = W * H
for i, data in enumerate(files):
    sg, lut = hip.png_parse(data, W, H)
    base = len(blob)
    zlen = sum(l for _, l in sg)
    frames_np[i] = (len(segs), len(sg), zoff, zlen, 0xFFFFFFFF, 0, (i * P) & 0xFFFFFFFF, (i * P) >> 32)
    segs += [(base + o, l) for o, l in sg]
    blob += data
    blob += b"\0" * ((-len(blob)) % 4)
    zoff += ((zlen + 15) & ~15) + 16
blob += b"\0" * 8
dev = torch.device("cuda:0")
d_files = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
d_frames = torch.from_numpy(frames_np.view(np.int32).copy()).to(dev)
d_segs = torch.tensor(segs, dtype=torch.int64).to(torch.int32).to(dev)
d_luts = torch.zeros(256, dtype=torch.uint8, device=dev)
stride = int(_lib.lib().abub_png_raw_stride(W, H))
d_z = torch.empty((zoff,), dtype=torch.uint8, device=dev)
d_raw = torch.empty((n * stride,), dtype=torch.uint8, device=dev)
out = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
status = torch.zeros((n,), dtype=torch.int32, device=dev)


def go():
    _lib.check(_lib.lib().abub_png_decode_dev(d_files.data_ptr(), d_files.numel(), d_frames.data_ptr(), n, d_segs.data_ptr(), len(segs),
                                              d_luts.data_ptr(), 0, W, H, d_z.data_ptr(), d_z.numel(), d_raw.data_ptr(), d_raw.numel(),
                                              out.data_ptr(), out.numel(), status.data_ptr(), torch.cuda.current_stream().cuda_stream), "png")


go()
torch.cuda.synchronize()
ts = []
for r in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = min(ts)
if os.environ.get("ABUB_PNG_DEBUG") == "5":
    stv = status.cpu().numpy().astype(np.int64)
    print("hand-over probe: mean pick-up latency (cycles, s_memtime) over the waits:", float(np.mean((-stv) % 1000000)), "waits per stream (thousands):",
          float(np.mean((-stv) // 1000000)))
print(f"{n} frames {W}x{H} level {level}: {ms:.2f} ms per batch = {n / ms * 1e3:.0f} frames/s, {n * P / ms / 1e6:.1f} GB/s of pixels; all: {[round(t, 2) for t in ts]}")

#!/usr/bin/env python3
"""Lists VGPR / SGPR / LDS / scratch of the gfx950 kernels inside a built .so (default: autobub3hs_amd/libabub_hip.so).
Usage: python3 tools/kernel_resources.py [lib.so] [name-regex]"""
import os, re, struct, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "autobub3hs_amd", "libabub_hip.so")
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
data = open(lib, "rb").read()
for n, m in enumerate(re.finditer(b"\x7fELF\x02\x01\x01\x40", data)):
    i = m.start()
    shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(data[i:i + shoff + shentsize * shnum])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    cur = {}
    rows = []
    for line in out.splitlines():
        line = line.strip()
        mm = re.match(r"-?\s*\.(\w+):\s*(.*)", line)
        if not mm:
            continue
        k, v = mm.groups()
        if k == "name" and "kd" not in v and v.startswith("_Z") or k == "name" and not v.startswith("'") and "(" not in v and cur.get("seen"):
            pass
        if k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "agpr_count"):
            cur[k] = v
        if k == "symbol":
            cur["symbol"] = v
        if k == "wavefront_size":
            rows.append(cur)
            cur = {}
    for r in rows:
        sym = r.get("symbol", "?").replace(".kd", "")
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", sym], capture_output=True, text=True).stdout.strip()
        except Exception:
            dem = sym
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        if pat and not pat.search(dem):
            continue
        print(f"{dem:60s} vgpr {r.get('vgpr_count','?'):>4s} agpr {r.get('agpr_count','-'):>3s} sgpr {r.get('sgpr_count','?'):>4s} "
              f"lds {r.get('group_segment_fixed_size','?'):>6s} scratch {r.get('private_segment_fixed_size','?'):>4s} spill {r.get('vgpr_spill_count','0')}")

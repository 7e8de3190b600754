#!/usr/bin/env python3
"""Disassembles one gfx950 kernel out of a built .so: python3 tools/disasm_kernel.py lib.so 'mangled-name-regex' > out.s"""
import os, re, subprocess, sys, tempfile
lib, pat = sys.argv[1], re.compile(sys.argv[2])
data = open(lib, "rb").read()
import struct
for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data):
    i = m.start()
    shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(data[i:i + shoff + shentsize * shnum])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    cur, buf = None, []
    for line in out.splitlines():
        mm = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if mm:
            if cur and pat.search(cur):
                print("\n".join(buf))
            cur, buf = mm.group(1), [line]
        else:
            buf.append(line)
    if cur and pat.search(cur):
        print("\n".join(buf))

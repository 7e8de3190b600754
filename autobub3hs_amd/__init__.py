"""autobub3hs_amd -- MI355X-native hot path of the PICO AutoBub3hs bubble finder.

csrc/        HIP kernels + C-ABI (include/abub_hip.h)  -> libabub_hip.so
_lib.py      ctypes loader (fails loudly if the HIP library is missing)
hip.py       device-pointer launchers for torch-owned HBM (tests, bench)
synth.py     seeded synthetic events (numpy == torch, integer only)
"""

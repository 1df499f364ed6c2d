#!/usr/bin/env python3
"""List VGPR / SGPR / scratch / LDS of every kernel in a hipcc -S --cuda-device-only listing:
   hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/engine.s smafa_amd/csrc/engine.hip
   python3 tools/kernel_resources.py /tmp/engine.s [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if name.endswith(".kd"): continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if want not in dem or "smafa::" not in dem: continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    print("%-62s vgpr %3s sgpr %3s scratch %4s lds %6s" % (dem.replace("void smafa::", ""), g("vgpr_count"), g("sgpr_count"),
          g("private_segment_fixed_size"), g("group_segment_fixed_size")))

#!/usr/bin/env python3
"""Register / scratch / LDS footprint of every kernel in a hipcc -S --cuda-device-only listing:
python tools/kernel_regs.py file.s [name-filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    f = {k: v for k, v in re.findall(r"\.(agpr_count|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|name|vgpr_spill_count):\s+(\S+)", "  - .agpr_count:" + blk)}
    if flt in f.get("name", ""):
        print(f"{f.get('name','?')[:90]:90s} vgpr {f.get('vgpr_count')} agpr {f.get('agpr_count')} scratch {f.get('private_segment_fixed_size')} spill {f.get('vgpr_spill_count')} lds {f.get('group_segment_fixed_size')}")

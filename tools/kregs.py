#!/usr/bin/env python
"""Print VGPR / AGPR / spill / LDS metadata per kernel from a hipcc -save-temps .s file (diagnostic)."""
import re, sys
t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in t.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        print(f"{name[:110]:110s} agpr={blk.split()[0]:>3s} vgpr={g('vgpr_count'):>3s} spill={g('vgpr_spill_count'):>3s} sgpr={g('sgpr_count'):>3s}")

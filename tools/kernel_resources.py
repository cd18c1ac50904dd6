"""Prints VGPR / scratch / occupancy / LDS of the kernels named on the command line from hipcc's
-Rpass-analysis=kernel-resource-usage output (files given first).  Design aid."""
import re
import sys

files = [a for a in sys.argv[1:] if a.endswith(".txt")]
pats = [a for a in sys.argv[1:] if not a.endswith(".txt")] or [""]
for f in files:
    txt = open(f).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        if not any(p in name for p in pats):
            continue
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        print(name[:70], "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
              "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))

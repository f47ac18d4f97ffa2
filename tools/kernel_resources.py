"""Per-kernel register/scratch/occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
Usage: python tools/kernel_resources.py <remarks file> [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split()[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if want not in d:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    short = re.sub(r"^void ", "", d.replace("(anonymous namespace)::", "").split("(")[0])
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{short:58s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} SGPR {g('SGPRs'):>3s} scratch {scr:>4s} occ {occ}")

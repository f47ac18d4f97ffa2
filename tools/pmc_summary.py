"""Summarise rocprofv3 --pmc passes (tools/pmc_sweep.sh) per kernel name and grid size: mean per dispatch."""
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "sweep"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    meta = {}
    for row in csv.DictReader(open(f)):
        if want not in row["Kernel_Name"]:
            continue
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per[key] += float(row["Counter_Value"])
        meta[row["Dispatch_Id"]] = (row["Kernel_Name"].split("(")[0][-60:], row.get("Grid_Size", "?"))
    # tools/run_sweep.py alternates K_nm.v and K_mn.u; when both use the same grid, tell them apart by order
    if "--alternate" in sys.argv:
        order = sorted(meta, key=int)
        meta = {d: (meta[d][0] + (" [knm]" if i % 2 == 0 else " [kmn]"), meta[d][1]) for i, d in enumerate(order)}
    for (did, cname), v in per.items():
        acc[meta[did]][cname].append(v)
out = {}
for k, d in acc.items():
    out[f"{k[0]} grid={k[1]}"] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    out[f"{k[0]} grid={k[1]}"]["dispatches"] = max(len(v) for v in d.values())
print(json.dumps(out, indent=1))

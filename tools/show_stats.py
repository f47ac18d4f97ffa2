"""Pretty-print a rocprofv3 *_kernel_stats.csv."""
import csv, glob, sys
pat = sys.argv[1]
f = sorted(glob.glob(pat))[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "::" in n and "(" in n:  # "void (anonymous namespace)::kernel<...>(args)" -> "kernel<...>"
        n = n.split("(anonymous namespace)::")[-1]
        depth, cut = 0, len(n)
        for i, ch in enumerate(n):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        n = n[:cut]
    n = n[:64]
    print(f'{n:66s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:10.1f} total_ms={float(r["TotalDurationNs"])/1e6:9.1f} {r["Percentage"]}%')

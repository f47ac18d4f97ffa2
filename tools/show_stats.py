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

# Optional second argument: the *_kernel_trace.csv of the same run.  The --stats table aggregates by kernel NAME; the
# fused sweep is launched at several sizes by one bench.py run (the 2^20-row CG steps, but also 4096-row prediction
# batches and the preconditioner's row sample), so its name-level average is not the average of the timed launches.
# This prints the sweep's launches grouped by grid size, which is.
if len(sys.argv) > 2:
    from collections import defaultdict
    g = defaultdict(list)
    for r in csv.DictReader(open(sorted(glob.glob(sys.argv[2]))[0])):
        if "sweep_fast_kernel" in r["Kernel_Name"] or "sweep_kernel" in r["Kernel_Name"]:
            g[(r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][:48], int(r.get("Grid_Size") or r["Grid_Size_X"]))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("\nfused sweep launches of this run by grid size (threads), from the kernel trace:")
    for (n, grid), d in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {n:48s} grid={grid:>9d} calls={len(d):>6d} avg_us={sum(d) / len(d):10.1f} total_ms={sum(d) / 1e3:9.1f}")

"""Pretty-print a rocprofv3 *_kernel_stats.csv."""
import csv, glob, sys
pat = sys.argv[1]
f = sorted(glob.glob(pat))[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0][-64:]
    print(f'{n:66s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:10.1f} total_ms={float(r["TotalDurationNs"])/1e6:9.1f} {r["Percentage"]}%')

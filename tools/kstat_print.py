"""Print a rocprofv3 kernel-stats CSV compactly: python tools/kstat_print.py <dir or csv> [substring ...]"""
import csv, glob, os, sys
path = sys.argv[1]
files = [path] if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))
for f in files:
    print(f)
    for r in csv.DictReader(open(f)):
        if len(sys.argv) > 2 and not any(k in r["Name"] for k in sys.argv[2:]):
            continue
        print("   %-72s calls=%-4s avg=%10.3f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))

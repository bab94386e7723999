"""Timeline of the last ws_search_host call in a trace written by tools/e2e_trace.py."""
import csv, sys
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(d + "/run_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for r in csv.DictReader(open(d + "/run_memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "")) ))
ev.sort()
# the last call = events after the last gap of more than 200 us
start = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[:i][-8:]) > 200000: start = i
t0 = ev[start][0]
for s, e, n in ev[start:]:
    print("%8.1f .. %8.1f us  (%6.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))

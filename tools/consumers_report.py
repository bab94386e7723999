"""Per-launch durations of the consumer kernels from a rocprofv3 kernel trace, grouped by grid size (= map size),
with the fraction of the 8 TB/s HBM peak their ALGORITHMIC bytes amount to:
  removeDisparityOutliers (ws_box_rows + ws_outlier_cols): 4 B in + 4 B out per pixel, both launches together;
  depth + vertices (ws_depth_vertices):                   4 B in (+3 B colour) + 4 B depth / 16 B position + 4 B colour out."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "ws_box_rows" in name or "ws_outlier" in name or "ws_depth_vertices" in name or "ws_depth4" in name:
        wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 0
        key = (name.split("(")[0].replace("void wsamd::", "")[:40], int(r["Grid_Size_X"]) * max(1, int(r.get("Grid_Size_Y", 1))))
        by[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for key in sorted(by):
    runs = [d for _, d in sorted(by[key])]
    parts = [("", runs)]
    if "ws_depth4" in key[0]:
        parts = [(" depth only (8 B/px, 4 pixels per thread: grid is not the pixel count)", runs)]
    elif "ws_depth_vertices" in key[0]:
        parts = [(" vertices+colours (27 B/px)", runs)]
    for tag, v in parts:
        v = sorted(v)
        print("%-42s grid %9d  launches %3d  median %8.2f us  min %8.2f us%s" % (key[0], key[1], len(v), v[len(v) // 2], v[0], tag))

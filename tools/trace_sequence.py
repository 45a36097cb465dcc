"""Print the dispatch sequence of a window of a rocprofv3 kernel trace (index, start offset, duration, gap to previous, kernel, grid).
usage: python tools/trace_sequence.py <dir> <first-from-end> <count>"""
import csv
import glob
import re
import sys

path, back, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
win = rows[len(rows) - back: len(rows) - back + count]
t0 = int(win[0]["Start_Timestamp"])
prev_end = None
for i, r in enumerate(win):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
    g = "%dx%dx%d" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]),
                      int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%5d %9.1f us  dur %7.1f  gap %6.1f  %-60s %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, gap, n, g))
    prev_end = e

"""Group the dispatches of a rocprofv3 kernel_trace CSV by (kernel, grid) - per-step counts and time.
usage: python tools/trace_by_grid.py <dir> <steps-equivalent> [name-filter] [top]"""
import csv
import glob
import re
import sys
from collections import defaultdict

path, steps = sys.argv[1], float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    wg = (int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"]))
    g = (int(r["Grid_Size_X"]) // wg[0], int(r["Grid_Size_Y"]) // wg[1], int(r["Grid_Size_Z"]) // wg[2])
    a = acc[(n[:70], g)]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in acc.values())
print("total %.3f ms/step over %d kinds" % (tot / steps / 1e3, len(rows)))
for (n, g), (c, us) in rows[:top]:
    print("%-72s grid %5d x %3d x %3d  calls/step %6.1f  avg %8.1f us  ms/step %7.3f" % (n, g[0], g[1], g[2], c / steps, us / c, us / steps / 1e3))

"""One steady-state training step out of a rocprofv3 kernel trace of `bench.py --step-only`: the dispatches between two
consecutive generator-optimizer launches (the second nadam_update_kernel of an iteration), in start order, with stream-agnostic
gaps, plus launch count / kernel time by kernel name for that step.
usage: python tools/step_sequence.py <dir> [steps-from-end=2] > sequence.txt"""
import collections
import csv
import glob
import re
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
nadam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("nadam_update_kernel")]
# two optimizer launches per iteration; the G optimizer is followed by its panel re-pack (two launches): cut behind those
ends = nadam[1::2]
lo, hi = ends[-back - 1] + 3, ends[-back] + 3
win = rows[lo:hi]
t0 = int(win[0]["Start_Timestamp"])
print("# one step: %d launches, %.3f ms from first start to last end" % (len(win), (max(int(r["End_Timestamp"]) for r in win) - t0) / 1e6))
by = collections.OrderedDict()
prev_end = None
for i, r in enumerate(win):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:56]
    g = "%dx%dx%d" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]),
                      int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%5d %9.1f us  dur %7.1f  gap %6.1f  q%-3s %-56s %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), n, g))
    prev_end = max(prev_end or 0, e)
    c = by.setdefault(n, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
print("# by kernel: launches, total us")
for n, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("# %5d %9.1f  %s" % (c, us, n))

"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV.  usage: pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if flt and flt not in n:
        continue
    key = (n[:60], r["Grid_Size"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
for key, d in acc.items():
    k = len(cnt[key])
    print("%s grid=%s dispatches=%d" % (key[0], key[1], k))
    for c, v in sorted(d.items()):
        print("    %-32s %.4g" % (c, v / k))

"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV.
usage: pmc_summary.py <dir> [name-filter]
       pmc_summary.py --json <out.json> <fetch_dir> <write_dir>
The --json form combines the FETCH_SIZE pass and the WRITE_SIZE pass of `tools/roofline_stages.py` (two separate rocprofv3
--pmc runs: the two counters do not fit one pass, MI355X_MICROARCH.md) into the record bench.py reads for `roofline.traffic`:
per stage, in launch order, the mean KiB per dispatch of the implicit-GEMM kernel, the HBM-side bytes per launch
2 * FETCH_SIZE + WRITE_SIZE (gfx950 counts a 128-byte read request as 64 bytes: the guide's correction), and the SHA-256 of
locate_amd/csrc/conv.hip at the time - bench.py reports the traffic only while that still matches the source it runs."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict, OrderedDict


def per_kernel(path, flt=""):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    acc = OrderedDict()
    cnt = defaultdict(set)
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if flt and flt not in n:
            continue
        key = (n[:60], r["Grid_Size"])
        acc.setdefault(key, defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key].add(r["Dispatch_Id"])
    return acc, cnt


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--json":
        out, fdir, wdir = sys.argv[2:5]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        src = os.path.join(root, "locate_amd", "csrc", "conv.hip")
        fetch, fc = per_kernel(fdir, "conv_igemm_bx6_kernel")
        write, wc = per_kernel(wdir, "conv_igemm_bx6_kernel")
        stages = []
        for key in fetch:
            if key not in write:
                continue
            f_kib = fetch[key]["FETCH_SIZE"] / len(fc[key])
            w_kib = write[key]["WRITE_SIZE"] / len(wc[key])
            stages.append({"kernel": key[0], "grid_threads": int(key[1]), "dispatches": len(fc[key]),
                           "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
                           "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024)})
        rec = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/roofline_stages.py",
               "correction": "HBM-side bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B)",
               "conv_hip_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(), "stages": stages}
        json.dump(rec, open(out, "w"), indent=1)
        print(json.dumps(rec, indent=1))
        return
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc, cnt = per_kernel(path, flt)
    for key, d in acc.items():
        k = len(cnt[key])
        print("%s grid=%s dispatches=%d" % (key[0], key[1], k))
        for c, v in sorted(d.items()):
            print("    %-32s %.4g" % (c, v / k))


if __name__ == "__main__":
    main()

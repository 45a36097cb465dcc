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


def per_dispatch(path, flt, counter):
    """[(kernel, grid, value)] of one counter in dispatch order."""
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    flts = flt if isinstance(flt, (tuple, list)) else (flt,)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and any(x in r["Kernel_Name"] for x in flts)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60], int(r["Grid_Size"]), float(r["Counter_Value"])) for r in rows]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--json":
        out, fdir, wdir = sys.argv[2:5]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from tools.roofline_stages import STAGES
        srcs = [os.path.join(root, "locate_amd", "csrc", n) for n in ("conv.hip", "convwin.hip", "igemm.h")]
        kernels = ("conv_igemm_bx6_kernel", "conv_win_kernel")          # whichever form each stage takes
        fetch = per_dispatch(fdir, kernels, "FETCH_SIZE")
        write = per_dispatch(wdir, kernels, "WRITE_SIZE")
        assert len(fetch) == len(write) and len(fetch) % len(STAGES) == 0, (len(fetch), len(write))
        per = len(fetch) // len(STAGES)                  # launches per stage (reps + 1), in stage order
        batch = 64
        stages = []
        for i, (c, size) in enumerate(STAGES):
            fs, ws = fetch[i * per:(i + 1) * per], write[i * per:(i + 1) * per]
            assert len({(k, g) for k, g, _ in fs}) == 1, fs
            f_kib = sum(v for _, _, v in fs) / per
            w_kib = sum(v for _, _, v in ws) / per
            out_bytes = 4 * batch * c * (2 * size) ** 2
            stages.append({"C": c, "in": size, "kernel": fs[0][0], "grid_threads": fs[0][1], "dispatches": per,
                           "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1), "output_bytes": out_bytes,
                           "hbm_bytes_per_launch": int(2.0 * f_kib * 1024 + w_kib * 1024)})
        rec = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- python3 tools/roofline_stages.py --reps 5",
               "correction": "HBM-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE.  FETCH_SIZE doubled: gfx950 tallies a 128-byte read "
                             "request as 64 bytes (MI355X_MICROARCH.md, HBM).  WRITE_SIZE as counted: round 3 read the un-split stages' "
                             "WRITE_SIZE = 2.000 x output bytes as a tally artefact of 4-byte-per-lane stores and subtracted one output; "
                             "round 4 shows it was REAL traffic - each sub-pixel phase wrote every other word of a line - because the same "
                             "stores count ~1.0 x the output once the four phases of a tile run as neighbours on one XCD (phase-fastest "
                             "tile order, igemm.h xcd_tile) and their stores meet in its L2.",
               "conv_hip_sha256": hashlib.sha256(b"".join(open(x, "rb").read() for x in srcs)).hexdigest(), "stages": stages}
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

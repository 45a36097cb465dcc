"""Summarise a rocprofv3 kernel_stats CSV of `bench.py --steps S --warmup W --no-graph` into per-step categories."""
import csv
import glob
import sys

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
cats = [("conv igemm", lambda n: "conv_igemm" in n or "conv_pointwise" in n or "conv_win" in n), ("conv wgrad", lambda n: "conv_wgrad" in n or "pw_wgrad" in n),
        ("conv 1x1 maps", lambda n: "skinny_" in n),
        ("slab reduce", lambda n: "slab_reduce" in n), ("panel pack", lambda n: "pack_" in n),
        ("spectral norm", lambda n: n.startswith("void sn_") or n.startswith("sn_")),
        ("norm", lambda n: "norm_" in n or "stats_" in n or "channel_sum" in n),
        ("roottanh/tanh", lambda n: "unary_" in n), ("gate", lambda n: "gate_" in n),
        ("resample/copy", lambda n: any(k in n for k in ("upsample", "avgpool", "feature_pool", "copy_planes"))),
        ("softmax", lambda n: "softmax" in n), ("nadam/loss", lambda n: "nadam" in n or "_loss_" in n),
        ("torch/runtime", lambda n: "at::native" in n or "rocclr" in n or "hip" in n.lower())]
tot = {c: [0, 0.0] for c, _ in cats}
tot["other"] = [0, 0.0]
for r in rows:
    n = r["Name"]
    for c, f in cats:
        if f(n):
            break
    else:
        c = "other"
    tot[c][0] += int(r["Calls"])
    tot[c][1] += float(r["TotalDurationNs"]) / 1e6
print("%-16s %12s %10s" % ("category", "launches/step", "ms/step"))
for c, (n, ms) in tot.items():
    print("%-16s %12.1f %10.3f" % (c, n / steps, ms / steps))
print("%-16s %12.1f %10.3f" % ("TOTAL", sum(v[0] for v in tot.values()) / steps, sum(v[1] for v in tot.values()) / steps))

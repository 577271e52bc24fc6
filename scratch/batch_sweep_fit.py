"""Fit t = a + b * rays per kernel over the rocprofv3 kernel-stats files scratch/batch_sweep.sh leaves (stats_<rays>.csv): a = what does not
shrink with the batch (zero fills, per-bin walks, capacity-sized loops, launch cost), b = the per-ray cost.  -> <dir>/batch_sweep.json + a table."""
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
steps = 25                                                  # 20 timed + 5 warm-up steps per run
per = {}
wall = {}
for f in sorted(glob.glob(os.path.join(d, "stats_*.csv")), key=lambda p: int(re.findall(r"stats_(\d+)", p)[0])):
    n = int(re.findall(r"stats_(\d+)", f)[0])
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").strip()
        calls = int(r["Calls"])
        if calls < steps:                                   # set-up kernels (cast, ray generation, fills)
            continue
        per.setdefault(name, {})[n] = calls * float(r["AverageNs"]) / 1e3 / steps       # microseconds per step (all launches of the kernel)
    log = os.path.join(d, f"n{n}.log")
    if os.path.exists(log):
        m = re.search(r'"ms_per_step": ([0-9.]+)', open(log).read())
        if m:
            wall[n] = float(m.group(1)) * 1e3
ns = sorted({n for v in per.values() for n in v})
out = {"_note": "microseconds per training step (sum over the kernel's launches) by rays per step; fit t = a + b * rays (least squares over the sizes present); "
                "intercept_frac = a / t(16384).  Source: rocprofv3 --kernel-trace --stats of `bench.py --task recon --rays N --steps 20 --warmup 5 --no-variants --no-roofline`.",
       "rays": ns, "kernels": {}, "step_wall_us": wall}


def fit(xs, ys):
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    sxx = sum((x - mx) ** 2 for x in xs)
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sxx if sxx else 0.0
    return my - b * mx, b


rows = []
for name, v in per.items():
    xs = [n for n in ns if n in v]
    if len(xs) < 2:
        continue
    a, b = fit(xs, [v[n] for n in xs])
    t16 = v.get(16384, a + b * 16384)
    out["kernels"][name] = {"us_per_step": {str(n): round(v[n], 2) for n in xs}, "a_us": round(a, 2), "b_us_per_kray": round(b * 1024, 3),
                            "intercept_frac": round(a / t16, 3) if t16 else None}
    rows.append((t16, name, v, a, b))
json.dump(out, open(os.path.join(d, "batch_sweep.json"), "w"), indent=1)
print("%-44s" % "kernel (us per step)" + "".join("%9d" % n for n in ns) + "   a(us)  b(us/kray)  a/t16k")
for t16, name, v, a, b in sorted(rows, reverse=True):
    print("%-44s" % name[:44] + "".join("%9.1f" % v.get(n, float('nan')) for n in ns) + "  %6.1f  %9.3f  %6.2f" % (a, b * 1024, a / t16 if t16 else 0))
print("%-44s" % "kernel sum" + "".join("%9.1f" % sum(v.get(n, 0) for _, _, v, _, _ in rows) for n in ns))
print("%-44s" % "step wall (bench, under the tracer)" + "".join("%9.1f" % wall.get(n, float('nan')) for n in ns))

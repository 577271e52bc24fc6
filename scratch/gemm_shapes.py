"""per-shape GEMM time of one eager edit step (which shapes carry the SDS half)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--task", "edit", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"]
import torch
from customnerf_amd.sd import ops as sdops
import bench
recs = []
orig = sdops.set_profile
def hook(r):
    orig(r)
    if r is not None: recs.append(r)
sdops.set_profile = hook
bench.main()
prof = recs[-1]
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, fl, shp in prof:
    a = acc[shp]; a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in acc.values())
print("total GEMM ms", tot, "launches", len(prof))
for shp, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("M%-7d N%-5d K%-6d mode%d b%-3d Hin%-4d Cin%-5d ts%d up%d  x%-3d %7.3f ms  %6.1f TF  %4.1f%%" % (*shp, a[0], a[1], a[2] / a[1] / 1e9, 100 * a[1] / tot))

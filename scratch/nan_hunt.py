"""Round 6: one bench run (r06a) had the edit leg's loss NaN on every step.  Repeat the edit leg a few times and report skipped steps / loss;
with CNERF_GRID_SPT / CNERF_GRID_TRAV (tuning build) the gather variant is forced.  Also checks the sample-major gather against the level-major
one on the edit leg's own sample lists (train and eval mode of the renderer)."""
import json, subprocess, sys, os
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for i in range(n):
    out = subprocess.run([sys.executable, "bench.py", "--task", "edit", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-variants", "--no-roofline"],
                         capture_output=True, text=True)
    try:
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        print(i, os.environ.get("CNERF_GRID_SPT"), os.environ.get("CNERF_GRID_TRAV"), "ms", round(d["ms_per_step"], 2), "loss", d["config"]["final_loss"], "skipped", d["config"]["steps_skipped_on_overflow"], d["config"]["loss_scale"][-12:], flush=True)
    except Exception as e:
        print(i, "FAILED", e, out.stderr[-500:], flush=True)

#!/bin/bash
# per-step kernel inventory of the timed edit step: kernel trace, window = the last 5 steps (between k_bin3_emit launches)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${1:-r05r}
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof_esk -o bench -- python3 bench.py --task edit --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $out/prof_esk.log 2>&1
python3 - <<E
import csv, glob, collections
f = glob.glob('$out/prof_esk/**/bench_kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
emit = [s for s, e, n in rows if n.startswith('k_bin3_emit')]
t0, t1 = emit[-6], emit[-1]
acc = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if t0 <= s < t1:
        acc[n][0] += 1; acc[n][1] += e - s
glue = {k: v for k, v in acc.items() if 'at::native' in k or '__amd_rocclr' in k}
tot = sum(v[1] for v in acc.values()) / 5e3
print(f"window: 5 steps, {(t1 - t0) / 5e6:.3f} ms/step wall, {tot:.1f} us/step kernel time, {sum(v[0] for v in acc.values()) / 5:.1f} launches/step")
print(f"ATen / rocclr: {sum(v[0] for v in glue.values()) / 5:.1f} launches/step, {sum(v[1] for v in glue.values()) / 5e3:.1f} us/step")
with open('$out/edit_step_kernels.txt', 'w') as fh:
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        line = f"{v[0] / 5:7.1f}/step {v[1] / 5e3:9.1f} us/step  {k[:150]}"
        fh.write(line + "\n")
        if k in glue: print(line)
E
rm -rf $out/prof_esk

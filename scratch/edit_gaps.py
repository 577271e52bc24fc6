"""GPU idle time inside an edit step: reads a rocprofv3 kernel trace (csv) of `bench.py --task edit`; a step = the interval between two consecutive
k_add_noise launches (one per step); reports, for the steps shorter than 30 ms (the timed graph-replay steps), busy union vs span and where the idle
time sits (by the kernel that follows the gap)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]) for r in rows)
marks = [i for i, e in enumerate(ev) if 'k_add_noise' in e[2]]
tot_span = tot_busy = 0
by = collections.Counter(); n_steps = 0
for a, b in zip(marks[:-1], marks[1:]):
    seg = ev[a:b]
    span = ev[b][0] - seg[0][0]
    if span > 30e6:
        continue
    n_steps += 1
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    for s, e, n in seg[1:] + [(ev[b][0], ev[b][0], ev[b][2])]:
        if s > cur_e:
            busy += cur_e - cur_s
            by[n] += s - cur_e
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot_span += span; tot_busy += busy
print(f"{n_steps} steps: span {tot_span/n_steps/1e6:.3f} ms, busy {tot_busy/n_steps/1e6:.3f} ms, idle {(tot_span-tot_busy)/n_steps/1e6:.3f} ms per step")
print("idle per step by the kernel that follows the gap:")
for n, g in by.most_common(22):
    print(f"  {g/n_steps/1e3:8.1f} us  {n}")

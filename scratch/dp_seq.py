import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]) for r in rows)
idx = [i for i, e in enumerate(ev) if 'k_dp_pack' in e[2]]
i0 = idx[-3]
t0 = ev[i0 - 6][0]
for s, e, n in ev[i0 - 6:i0 + 40]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n}")

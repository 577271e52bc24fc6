#!/bin/bash
# A/B of tuning-build switches on the recon leg, read from bench.py's own record (step time, gather coarse / fine launch times, stage events):
#   scratch/ab_recon.sh <tag> "<bench args>" "VAR=val VAR2=val" "VAR=val" ...      ("-" = no variables)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; bargs=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
i=0
for vars in "$@"; do
  i=$((i+1))
  [ "$vars" = "-" ] && vars=""
  env $vars timeout 300 python3 bench.py --task recon --steps 20 --warmup 5 --no-cpu-baseline --no-variants --stage-events $bargs > $out/ab_$i.json 2> $out/ab_$i.err
  python3 - "$out/ab_$i.json" "${vars:-base}" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    st = d["config"].get("stage_ms") or {}
    print("%-44s step %.3f ms | gather coarse %.1f fine %.1f us (%s) | emit %.0f accum %.0f split %.0f fieldbwd %.0f us" % (
        sys.argv[2], d["ms_per_step"], r.get("coarse", {}).get("avg_launch_ms", 0) * 1e3, r.get("fine", {}).get("avg_launch_ms", 0) * 1e3,
        (r.get("fine_traversal") or {}).get("choice"), (st.get("scatter_emit") or 0) * 1e3, (st.get("scatter_accumulate") or 0) * 1e3,
        (st.get("scatter_split_reduce") or 0) * 1e3, (st.get("field_backward") or 0) * 1e3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done

#!/bin/bash
# same-box A/B of a tuning switch on the edit leg: scratch/ab_env.sh VAR "v0 v1 ..." [bench args]   (library built with `make TUNING=1`)
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; shift; shift
for rep in 1 2; do
  for v in $vals; do
    r=$(env $var=$v python bench.py --task edit --no-cpu-baseline --no-variants --no-roofline "$@" 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print(round(r['ms_per_step'],3))")
    echo "$var=$v rep $rep: $r ms/step"
  done
done

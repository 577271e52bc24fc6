#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 2 0; do
CNERF_FIELD_X4_BWD=$v scratch/prof.sh r2prof_x$v 2>&1 | grep "ms_per_step\|k_field"
done

cd $GRAFT_REPO_ROOT
for v in "A=1" "CNERF_B2_STAGED=0" "CNERF_B2_INTACC=0" "CNERF_B2_PTS=4096"; do
echo "== $v"; env $v python -m pytest tests/test_gpu_gridencoder.py -x -q -m gpu -k "binned" 2>&1 | grep -E "passed|failed|Mismatched|Max abs" 
done

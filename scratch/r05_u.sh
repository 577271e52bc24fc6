#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05u
rm -rf /tmp/prev && mkdir -p /tmp/prev && tar -xf scratch/prev_tree.tar -C /tmp/prev && make -s -C /tmp/prev/customnerf_amd/csrc -j48 > gpurun_out/r05u/make_prev.log 2>&1
cp scratch/edit_host.py /tmp/prev/scratch/
echo "--- prev"; (cd /tmp/prev && timeout 300 python scratch/edit_host.py 2>&1 | grep -E "wall|isolated")
echo "--- new"; timeout 300 python scratch/edit_host.py 2>&1 | grep -E "wall|isolated"
echo "--- prev"; (cd /tmp/prev && timeout 300 python scratch/edit_host.py 2>&1 | grep -E "wall|isolated")

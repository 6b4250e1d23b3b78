#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lf7; mkdir -p $out
bash tools/prof_leaf.sh run > $out/prof.txt 2>&1
tail -18 $out/prof.txt

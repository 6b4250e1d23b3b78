#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lf1; mkdir -p $out
for a in "tiny" "tiny spec" "200k spec"; do
  echo "== $a"; timeout 300 python3 tools/lf_check.py $a 2>&1 | grep -v "^level" | tail -12
done > $out/check.log 2>&1
cat $out/check.log

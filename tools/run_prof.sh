#!/bin/bash
# per-workgroup timeline of the one-launch factor region (profile build made by tools/prof_factor.sh build)
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-prof}; mkdir -p $out
shift
DLG_FL_DUMP_ALL=1 "$@" bash tools/prof_factor.sh run > $out/raw.txt 2>&1
python3 tools/pr_timeline.py $out/raw.txt | tee $out/levels.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3e; mkdir -p $out
for cfg in "4 128 128" "4 128 192" "4 128 256" "4 128 384" "8 128 256" "4 64 256"; do
  set -- $cfg
  echo "replicas $1 fill0 $2 fill $3"
  DOGLEG_AMD_FRONT_REPLICAS=$1 DOGLEG_AMD_FRONT_FILL0=$2 DOGLEG_AMD_FRONT_FILL=$3 timeout 300 python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > $out/b_$1_$2_$3.json 2> $out/b_$1_$2_$3.err
  python3 tools/pj.py < $out/b_$1_$2_$3.json
done

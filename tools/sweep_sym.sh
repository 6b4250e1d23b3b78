#!/bin/bash
# tools only: knobs of the symbolic phase against steps/s (config #4)
cd "$(dirname "$0")/.." || exit 1
export DOGLEG_AMD_NO_SYM_CACHE=1
bash tools/sweep_env.sh DOGLEG_AMD_ND_LEAF 200 300 400 600
unset DOGLEG_AMD_ND_LEAF
bash tools/sweep_env.sh DOGLEG_AMD_RELAX_PCT 10 25 40 60
unset DOGLEG_AMD_RELAX_PCT
bash tools/sweep_env.sh DOGLEG_AMD_SIB_W 48 64 96
unset DOGLEG_AMD_SIB_W
bash tools/sweep_env.sh DOGLEG_AMD_SPLIT_W 16 32 64
unset DOGLEG_AMD_SPLIT_W
bash tools/sweep_env.sh DOGLEG_AMD_PERSIST_MAX 128 256 384

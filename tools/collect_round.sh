#!/bin/bash
# Collect the per-round artifacts on the GPU box: bench JSON lines, rocprofv3 kernel stats,
# PMC traffic (separate passes), end-to-end rate, probes.  usage: tools/collect_round.sh <tag>
# Writes under gpurun_out/<tag>/ ; tools/publish_round.py copies the summaries into profiles/.
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
tag=${1:-r01}; out=gpurun_out/$tag; mkdir -p $out
for wl in sparse-1m sparse-200k dense-50k; do
  timeout 600 python3 bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
done
timeout 900 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --cpu-seconds 5 > $out/bench_sparse-5m.json 2> $out/bench_sparse-5m.err
for wl in sparse-1m dense-50k; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o p -- python3 bench.py --workload $wl --no-cpu-baseline > $out/stats_$wl.log 2>&1
done
for wl in sparse-1m dense-50k; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${wl}_$c -o p -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_${wl}_$c.log 2>&1
    python3 tools/pmc_kernel.py $out/pmc_${wl}_$c > $out/pmc_${wl}_$c.txt
  done
done
DOGLEG_AMD_TIMING=1 timeout 600 python3 tools/e2e_bench.py --workload sparse-1m > $out/e2e_sparse1m.json 2> $out/e2e.err
# SQ counters of the assembly kernel (three passes of eight), the per-workgroup timeline of the one-launch factor region
bash tools/run_sq.sh $tag/sq > $out/sq.log 2>&1
cp gpurun_out/$tag/sq/sq_k4.txt $out/sq_k4.txt 2>/dev/null
# round 6: the same counters for the dense JtJ kernel k_syrk_lower<64> (78 % of config #2's step)
bash tools/run_sq_dense.sh $tag/sqd > $out/sqd.log 2>&1
cp gpurun_out/$tag/sqd/sq_syrk.txt $out/sq_syrk.txt 2>/dev/null
bash tools/run_prof.sh $tag/prof env > $out/top_of_tree_levels.txt 2>&1
mkdir -p $out/prof5; DLG_FL_DUMP_ALL=1 DLG_FL_DUMP_N=32 bash tools/prof_factor.sh run --workload sparse-5m > $out/prof5/raw.txt 2>&1; python3 tools/pr_timeline.py $out/prof5/raw.txt > $out/top_of_tree_levels_config5.txt 2>&1
python3 tools/k4_split.py > $out/k4_split.txt 2>&1; DLG_ASM_ONLY_SHAPE=0 python3 tools/k4_split.py >> $out/k4_split.txt 2>&1; DLG_ASM_ONLY_SHAPE=1 python3 tools/k4_split.py >> $out/k4_split.txt 2>&1
timeout 300 python3 tools/gpu_probe.py > $out/probe.txt 2>&1
# round 5: RCCL's cost per all-reduce at world size 1 (a process of its own), the dense triangular solves' hop timeline, the
# panel sweeps with and without the vector finish of a short last block, where workgroups land (XCDs) and what a hand-off costs
timeout 300 python3 tools/rccl_floor.py 2> $out/rccl_floor.err | grep '^{' > $out/rccl_floor.json
if [ -f tools/micro/libtrsvprof.so ]; then DLG_PROF_LIB=tools/micro/libtrsvprof.so timeout 300 python3 tools/trsv_prof.py 2>&1 | grep "trsv wg" > $out/trsv_hops.txt; fi
if [ -f tools/micro/libpotrfprof.so ]; then DLG_PROF_LIB=tools/micro/libpotrfprof.so timeout 300 python3 tools/potrf_prof.py 2>&1 | grep "potrf diag" > $out/potrf_diag.txt; fi
{
  echo "# tools/micro/bench_panel (one workgroup of 512 threads, panel in LDS; us per launch include ~5 us of load / store); second line of a pair: -DDLG_PF_NO_VFIN (the short last block on the matrix cores)"
  for a in "187 60" "193 66" "199 66" "205 66" "211 72" "127 126" "100 66" "163 36"; do echo "== nrows w = $a"; timeout 60 tools/micro/bench_panel 1 $a 512 | grep -E "B16 [0-9]|vfin phases"; timeout 60 tools/micro/bench_panel_novfin 1 $a 512 | grep -E "B16 [0-9]"; done
} > $out/panel_sweep.txt 2>&1
timeout 120 tools/micro/xcd_probe > $out/xcd_probe.txt 2>&1
DLG_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python3 bench.py --no-cpu-baseline > $out/bench_dist_world1_rccl.log 2>&1
timeout 1500 python3 tools/scaling_projection.py --workload sparse-1m --ranks 8 --rccl-floor $out/rccl_floor.json > $out/scaling_projection.md 2> $out/scaling_projection.err
timeout 2400 python3 tools/scaling_projection.py --workload sparse-5m --ranks 8 --steps 8 --lambda0 1.0 > $out/scaling_projection_sparse5m.md 2>> $out/scaling_projection.err
bash tools/run_trace.sh $tag/trace > $out/step_trace.txt 2>&1
# keep the merge small: only summaries travel back
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
find $out -name "*agent_info.csv" -delete; find $out -name "*domain_stats.csv" -delete
ls -la $out

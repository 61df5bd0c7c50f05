#!/bin/bash
# Everything profiles/rNN/ holds for one round, produced ON the GPU box from the repo root:
#   bash tools/profile_round.sh r02        (writes gpurun_out/r02/, copy what is to be judged into profiles/r02/)
# 1. the bench line of every BASELINE workload (same box),  2. rocprofv3 --kernel-trace --stats of `python3 bench.py`
# for NS, C3, C5 (+ the bench line printed under the profiler),  3. HBM-side traffic of NS and C3 from separate --pmc
# passes (tools/pmc_traffic.py),  4. the per-axis CPML cost A/B (tools/kernel_ab.py),  5. the two schedules on C3-C5.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r02}
mkdir -p $OUT
cd $R
for w in NS C2 C3 C4 C5; do
  extra="--no-hbm-point --no-cpu-baseline"; [ $w = NS ] && extra=""
  timeout -k 10 300 python3 bench.py --workload $w $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "bench $w failed" >&2
  echo "bench $w done"
done
cd /tmp && export TMPDIR=/tmp
for w in NS C3 C5; do
  rm -rf $OUT/trace_$w
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-point > $OUT/bench_${w}_under_rocprofv3.json 2> $OUT/trace_$w.err || echo "trace $w failed" >&2
  python3 $R/tools/rocprof_db_stats.py $(dirname $(ls $OUT/trace_$w/*/*.db | head -1)) > $OUT/kernel_stats_$w.csv
  rm -rf $OUT/trace_$w
  echo "trace $w done"
done
cd $R
for w in NS C3 C5; do
  timeout -k 10 600 python3 tools/pmc_traffic.py $w $OUT/pmc_traffic_${w}.json > $OUT/pmc_$w.log 2>&1 || echo "pmc $w failed" >&2
  rm -rf gpurun_out/pmc_${w}_FETCH_SIZE gpurun_out/pmc_${w}_WRITE_SIZE
  echo "pmc $w done"
done
timeout -k 10 300 python3 tools/kernel_ab.py NS,C3 CPML,PEC,xCPML,yCPML,zCPML 400 > $OUT/cpml_axis_cost.txt 2>&1
# 5. same-box A/B of the two schedules on the grids beyond the Infinity Cache: two launches per timestep (flags 1) vs one (flags 5)
for w in C3 C4 C5; do
  AB_FLAGS=1 AB_TAG=two_launches timeout -k 10 200 python3 tools/kernel_ab.py $w CPML,PEC 400 >> $OUT/wavefront_ab.txt 2>&1
  AB_FLAGS=5 AB_TAG=one_launch timeout -k 10 200 python3 tools/kernel_ab.py $w CPML,PEC 400 >> $OUT/wavefront_ab.txt 2>&1
done
timeout -k 10 200 python3 tools/plugin_path_timing.py > $OUT/plugin_path_timing_fixed_scene.txt 2>&1
lscpu | head -20 > $OUT/host_cpu.txt
echo "all done"

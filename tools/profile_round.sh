#!/bin/bash
# tools/profile_round.sh <tag>: everything profiles/<tag>/ holds that comes from one box — run ON a GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03'
# Writes under gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/).
set -u
tag=${1:-r04}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
# 1. bench lines of every BASELINE workload (the default line first: NS with the C3 point and the CPU baseline)
python3 bench.py > $out/bench_NS.json 2> $out/bench_NS.err
for w in C2 C3 C4 C5; do
  python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-point --no-small-grid-point --prefill-seconds 1 > $out/bench_$w.json 2> $out/bench_$w.err
done
echo "bench lines done" >&2
# 2. rocprofv3 kernel traces of the same command (program directly after --)
for w in NS C3 C5; do
  d=$out/trace_$w
  rm -rf $d
  if [ $w = NS ]; then args="--no-cpu-baseline --no-hbm-point --no-small-grid-point"; else args="--workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-point --no-small-grid-point --prefill-seconds 1"; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -- python3 $root/bench.py $args > $out/bench_${w}_under_rocprofv3.json 2> $out/trace_$w.err)
  python3 tools/rocprof_db_stats.py $(dirname $(find $d -name "*.db" | head -1)) > $out/kernel_stats_$w.csv 2>> $out/trace_$w.err
  rm -rf $d
  echo "trace $w done" >&2
done
# 3. PMC traffic (separate passes per counter, tools/pmc_traffic.py)
for w in NS C2 C3; do
  python3 tools/pmc_traffic.py $w $out/pmc_traffic_$w.json > /dev/null 2> $out/pmc_$w.err
  rm -rf gpurun_out/pmc_${w}_FETCH_SIZE gpurun_out/pmc_${w}_WRITE_SIZE
  echo "pmc $w done" >&2
done
# 4. per-axis CPML cost and the XCD-share A/B
python3 tools/kernel_ab.py NS,C3 CPML,PEC,xCPML,yCPML,zCPML 1000 > $out/cpml_axis_cost.txt 2>&1
lscpu | head -20 > $out/host_cpu.txt
echo "profile round $tag done" >&2

#!/bin/bash
# tools/mur_scene_trace.sh <tag>: rocprofv3 kernel trace of the reference GUI's default scene (56x55x50, MUR, to -40 dB) through the
# plugin path: per-kernel durations (--stats view) and the idle time between dispatches (tools/rocprof_db_gaps.py).
#   gpurun --timeout 600 -- 'bash tools/mur_scene_trace.sh r04'
set -u
tag=${1:-r04}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 tools/plugin_path_timing.py > $out/plugin_path_timing_fixed_scene.txt 2>&1
d=$out/trace_mur
rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -- python3 $root/tools/plugin_path_timing.py > $out/plugin_path_timing_under_rocprofv3.txt 2> $out/trace_mur.err)
dbdir=$(dirname $(find $d -name "*.db" | head -1))
python3 tools/rocprof_db_stats.py $dbdir > $out/mur_scene_kernel_stats.csv 2>> $out/trace_mur.err
python3 tools/rocprof_db_gaps.py $dbdir > $out/mur_scene_kernel_gaps.txt 2>> $out/trace_mur.err
rm -rf $d
echo "mur scene trace $tag done" >&2

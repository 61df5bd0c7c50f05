#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: builds csrc with extra flags into scratch/v/<name>/libfdtd_hip.so
# (scratch/ is git-ignored but travels to the GPU box); select it with FDTD_HIP_LIB_DIR=$PWD/scratch/v/<name>.
# e.g. tools/build_variant.sh trace -DFDTD_XCD_TRACE   (tools/xcd_trace.py);  tools/build_variant.sh b128 -DFDTD_BLOCK=128 -DFDTD_WF_MINBLOCKS=14 ...
set -e
name=$1; shift
src=/root/repo/fdtd-solver-antennas_amd/csrc
out=/root/repo/scratch/v/$name
mkdir -p $out
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -Wno-unused-result -w $*"
for f in api kernels resident opbuild farfield; do
  /opt/rocm/bin/hipcc $FL -I$src -c $src/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 $out/*.o -shared -Wl,-Bsymbolic -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $out/libfdtd_hip.so
rm -f $out/*.o
echo built $out

#!/bin/bash
# builds a variant of libwalnuts_hip.so with extra -D flags into tests/gpu_probes/libwalnuts_<name>.so
# usage: build_variant.sh <name> "<flags>"
set -e
HERE=$(cd $(dirname $0) && pwd); CS=$HERE/../../walnuts_amd/csrc; D=/tmp/wn_variant_$1; mkdir -p $D
rm -f $D/*.o
for f in wn_engine wn_sample wn_summary $(cd $CS && ls wn_kernels_*.hip | sed s/.hip//); do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -I$CS -DWN_FAST_BUILD ${VARIANT_BASE_FLAGS--mllvm -structurizecfg-skip-uniform-regions=1} $2 -c $CS/$f.hip -o $D/$f.o &
done; wait
for f in wn_engine wn_sample wn_summary $(cd $CS && ls wn_kernels_*.hip | sed s/.hip//); do
  [ -s $D/$f.o ] || { echo "build_variant: $f failed to compile" >&2; exit 1; }
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/libwalnuts_$1.so $D/*.o
echo built $HERE/libwalnuts_$1.so

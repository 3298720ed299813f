#!/usr/bin/env bash
# Dev: build a variant of the library with extra compiler flags into build_ab/lib_<name>.so (A/B runs: scripts/gpu_lib_ab.sh).
# usage: build_variant.sh <name> "<extra flags>" [make variable assignments, e.g. FLAGS_raster_g16=-fno-slp-vectorize]
set -e
name=$1; extra=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/gsl_variant_$name
rm -rf $tmp; mkdir -p $tmp/gsplatloc_amd $tmp/include $root/build_ab
cp -r $root/gsplatloc_amd/csrc $tmp/gsplatloc_amd/csrc
cp $root/include/*.h $tmp/include/
rm -f $tmp/gsplatloc_amd/csrc/*.o
make -C $tmp/gsplatloc_amd/csrc EXTRA="$extra" "$@" -j4 > $tmp/build.log 2>&1 || { grep -E "error" -A5 $tmp/build.log | head -20; exit 1; }
cp $tmp/gsplatloc_amd/libgsloc_hip.so $root/build_ab/lib_$name.so
echo built build_ab/lib_$name.so

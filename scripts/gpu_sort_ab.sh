#!/usr/bin/env bash
# Dev: tile-sort kernels A/B (GSL_DEV_TILE_SORT=wave|wg|unset = library's choice): the strip estimates at R and X.
set -o pipefail
for k in wave wg auto; do
  for wl in R X; do
    echo "== $k $wl"
    if [ $k = auto ]; then unset GSL_DEV_TILE_SORT; else export GSL_DEV_TILE_SORT=$k; fi
    timeout -k 10 600 python3 scripts/strip_scaling.py 1.0 random $wl 2>&1 | grep -a "^world\|world 8 rank"
  done
done

#!/bin/bash
# usage: tools/ab_distinct.sh <libA.so> <libB.so> ...  -- the dense uniqueness pass of the bench table's two key columns
# (tools/bench_distinct.py --dense-only, 1 G rows) under each build of the library, kernel times from rocprofv3
for lib in "$@"; do
  tag=ab_$(basename $lib .so)
  echo "=== $lib"
  TGX_LIB=$PWD/$lib tools/prof_any.sh $tag tools/bench_distinct.py --dense-only --steps 5 2>&1 | grep -E "^col|partition_kernel|bucket_apply" | grep -v "calls    1[05] *avg    0.00"
done

#!/bin/bash
# usage: build_variant.sh <name> [-DFLAG ...]   -> parquet2_amd/lib/variants/libpq2hip_<name>.so (A/B builds, loaded via PQ2_LIB)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $root/parquet2_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall "$@" \
  -o $root/parquet2_amd/lib/variants/libpq2hip_$name.so \
  $root/parquet2_amd/csrc/pq2_kernels.hip $root/parquet2_amd/csrc/pq2_noalloc.hip $root/parquet2_amd/csrc/pq2_api.cpp $root/parquet2_amd/csrc/pq2_host.cpp

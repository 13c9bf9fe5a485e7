#!/bin/bash
# usage: build_variant.sh <name> [-DFLAG ...]   -> parquet2_amd/lib/variants/libpq2hip_<name>.so (A/B builds, loaded via PQ2_LIB)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $root && python3 -m parquet2_amd.build --variant=$name "$@"

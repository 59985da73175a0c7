#!/bin/bash
# tools/diag/build_variant.sh <name> <extra hipcc flags...>  ->  gnn-motion-planning_amd/libgnnmp_<name>.so (GNNMP_LIB selects it)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
N=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -fPIC -shared "$@" -o $R/gnn-motion-planning_amd/libgnnmp_$N.so $R/gnn-motion-planning_amd/csrc/*.cpp $R/gnn-motion-planning_amd/csrc/*.hip

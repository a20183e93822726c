#!/bin/bash
# builds the boundary-cost probe next to this script (binaries are git-ignored)
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 aql_kernels.hip -o aql_kernels.hsaco
hipcc -O2 -std=c++17 aql_probe.cpp -o aql_probe -L/opt/rocm/lib -lhsa-runtime64

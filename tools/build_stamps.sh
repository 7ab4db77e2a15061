#!/bin/bash
# diagnostic build with per-phase cycle stamps (never used by the product path)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DNOCF_STAMPS -Iinclude -Ineuraloc_amd/csrc \
  -o neuraloc_amd/csrc/libnocf_stamps.so neuraloc_amd/csrc/nocf_kernels.hip

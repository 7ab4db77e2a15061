#!/bin/bash
# diagnostic build (never used by the product path): NOCF_STAMPS_LEVEL=1 (default) per-wave timeline of one evaluation
# (tools/timeline.py, tools/duo_timeline.py); =2 also the per-phase cycle accumulators of tools/phase_stamps.py (24 more VGPRs: distorts the kernel)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DNOCF_STAMPS=${NOCF_STAMPS_LEVEL:-1} -Iinclude -Ineuraloc_amd/csrc \
  -o neuraloc_amd/csrc/libnocf_stamps.so neuraloc_amd/csrc/nocf_kernels.hip neuraloc_amd/csrc/nocf_duo.hip

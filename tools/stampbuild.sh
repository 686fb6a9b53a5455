#!/bin/bash
# diagnostic build with in-kernel cycle stamps (prints per-section cycles); writes pde_amd/libpinnjet.so
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPINN_WITH_MFMA -DPINN_DEV_KDV7_ONLY -DPINN_STAMPS "$@" \
  pde_amd/csrc/pinnjet.hip -o pde_amd/libpinnjet.so 2>&1 | grep -E "error" | head

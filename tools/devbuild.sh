#!/bin/bash
# fast iteration build: only the kdv7 / NT=4 MFMA instantiations; prints register usage
cd "$(dirname "$0")/.."
time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPINN_WITH_MFMA -DPINN_DEV_KDV7_ONLY \
  pde_amd/csrc/pinnjet.hip -o pde_amd/libpinnjet.so -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|k_mfma|VGPRs|AGPRs|ScratchSize|Spill" | grep -A5 "k_mfma" | sed 's/.*remark: *//; s/ \[-Rpass.*//'

#!/bin/bash
# timing-ablation builds (kdv7 only, results are wrong with any PINN_ABL_* on, timing only):
#   tools/ablate.sh name "-DPINN_ABL_WATOM=1 ..."   ->  build_dev/lib_<name>.so
cd "$(dirname "$0")/.."
mkdir -p build_dev
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPINN_WITH_MFMA -DPINN_DEV_KDV7_ONLY -fno-slp-vectorize -w $@ \
  pde_amd/csrc/pinnjet.hip pde_amd/csrc/lbfgs.hip -o build_dev/lib_$name.so -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|k_mfmaILi3ELi50ELi2|VGPRs:|AGPRs|ScratchSize|Spill" | grep -A5 "k_mfmaILi3ELi50ELi2" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '
echo " [$name]"

#!/bin/bash
# as tools/ablate.sh, for the width-100 GNSS instantiation:  tools/ablate_gnss.sh name "<flags>" -> build_dev/lib_<name>.so
cd "$(dirname "$0")/.."
mkdir -p build_dev
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPINN_WITH_MFMA -DPINN_DEV_GNSS100_ONLY -fno-slp-vectorize -w $@ \
  pde_amd/csrc/pinnjet.hip pde_amd/csrc/lbfgs.hip -o build_dev/lib_$name.so -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|k_mfmaILi5ELi100ELi2|VGPRs:|AGPRs|ScratchSize|Spill" | grep -A5 "k_mfmaILi5ELi100ELi2" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '
echo " [$name]"

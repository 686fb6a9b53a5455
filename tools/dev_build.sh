#!/bin/bash
# Development build of ONE kernel family for quick iteration (not the shipped library):
#   tools/dev_build.sh <name> [extra hipcc flags]     -> build_dev/lib_<name>.so  (+ the ISA in build_dev/<name>/*.s)
# WHICH=KDV7 (default) | GNSS100 | CH selects the PINN_DEV_*_ONLY instantiation set of kernel_mfma.hpp.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_dev/$name
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DPINN_WITH_MFMA -DPINN_DEV_${WHICH:-KDV7}_ONLY"
/opt/rocm/bin/hipcc $flags "$@" -save-temps=obj -c pde_amd/csrc/pinnjet.hip -o build_dev/$name/pinnjet.o
[ -f build_obj/lbfgs.hip.o ] || /opt/rocm/bin/hipcc $flags -c pde_amd/csrc/lbfgs.hip -o build_obj/lbfgs.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_dev/$name/pinnjet.o build_obj/lbfgs.hip.o build_obj/adam.hip.o build_obj/dataloss.hip.o -o build_dev/lib_$name.so
s=$(ls build_dev/$name/*gfx950*.s | head -1)
grep -A40 "k_mfmaILi3ELi50ELi2E" $s | grep -m1 -A30 "\.amdhsa_kernel\|; Kernel info" >/dev/null 2>&1 || true
python3 - "$s" <<'PY'
import re,sys
t=open(sys.argv[1]).read()
for m in re.finditer(r"; Function info:|\.amdhsa_kernel (\S+)", t):
    pass
for blk in re.split(r"\n\s*\.amdhsa_kernel ", t)[1:]:
    name=blk.split("\n")[0]
    if "k_mfma" not in name: continue
    if not name.rstrip().endswith("ELi2EEvNS_5KDescENS_8CallArgsEPf") and "Li2EEv" not in name: continue
    g=lambda k: (re.search(r"\.amdhsa_%s (\S+)"%k, blk) or [None,"?"])[1]
    print(name[:60], "next_free_vgpr", g("next_free_vgpr"), "accum_offset", g("accum_offset"), "scratch", g("private_segment_fixed_size"))
PY

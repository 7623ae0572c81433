#!/bin/bash
# Builds variants of the library that differ only in how hip/pt_kernel.hip (the A/B megakernel) is compiled,
# under raytracer_2022_amd/variants/ — then, on the GPU box, `tools/mega_bisect.sh run` runs tools/dbg_mega2.py
# against each (RT2022_LIB selects the build). Evidence for the -O1 / -O3 divergence of the megakernel.
set -e
cd "$(dirname "$0")/.."
V=raytracer_2022_amd/variants
if [ "$1" = run ]; then
  for so in $V/mega_*.so; do
    echo "== $(basename $so .so)"
    RT2022_LIB=$PWD/$so timeout -k 10 300 python3 tools/dbg_mega2.py 2>&1 | tail -12
  done
  exit 0
fi
mkdir -p $V
build() {  # name, MEGA_OPT, MEGA_EXTRA
  make -s -j4 -C raytracer_2022_amd/csrc BUILD=../../build_mega_$1 OUT=../variants/mega_$1.so MEGA_OPT="$2" MEGA_EXTRA="$3"
  echo built mega_$1
}
build O1 -O1 ""
build O2 -O2 ""
build O3 -O3 ""
build O3_packed -O3 "-DRT2022_CHAIN_PACKED=1"
build O3_sgprmem -O3 "-mllvm -amdgpu-spill-sgpr-to-vgpr=0"
build O3_noslp -O3 "-fno-slp-vectorize -fno-vectorize"

#!/bin/bash
# diagnostic build of the library with in-kernel stamps in gemm_pp_kernel (never shipped, never timed as a kernel)
set -e
cd "$(dirname "$0")/../../gm-diffusion_amd/csrc"
make -j4 >/dev/null
mkdir -p build/diag
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -DGMD_PP_DIAG=1 -c gemm.hip -o build/diag/gemm.o
OBJS=$(ls build/*.o | grep -v "build/gemm.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/dbg/libgmd_ppdiag.so $OBJS build/diag/gemm.o
echo built tools/dbg/libgmd_ppdiag.so

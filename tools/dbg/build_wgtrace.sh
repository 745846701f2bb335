#!/bin/bash
# diagnostic build of the whole library with the per-workgroup occupancy trace (csrc/wg_trace.h); never shipped, never benchmarked.
# build_wgtrace.sh [extra -D flags]   e.g. OUT=libgmd_wgtrace_pp.so build_wgtrace.sh -DGMD_PP_DIAG=1  (in-kernel stamps of gemm_pp_kernel as detail records)
set -e
cd "$(dirname "$0")/../../gm-diffusion_amd/csrc"
OUT=${OUT:-libgmd_wgtrace.so}
BD=build/wgtrace_${OUT%.so}
mkdir -p $BD
FLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -DGMD_WG_TRACE=1 $*"
pids=()
for f in hdr_tail latent_step norm elementwise gemm gemm_split ff_fused attention attention_split; do
  extra=""; case $f in hdr_tail|latent_step) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc $FLAGS $extra -c $f.hip -o $BD/$f.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
make build/gmd_error.o build/rgbe_rle.o >/dev/null
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/dbg/$OUT $BD/*.o build/gmd_error.o build/rgbe_rle.o
echo built tools/dbg/$OUT

#!/bin/bash
# Debug build of the library with per-item cycle traces compiled in (-DLSR_ENABLE_TRACE):
#   build_variants/liblsr_trace.so   (select it with LSR_LIB=...; LSR_TRACE=<file> dumps the forward compositing items)
set -e
cd "$(dirname "$0")/../latentsplat_amd/csrc"
OUT=../../build_variants/trace_obj
mkdir -p $OUT
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize -DLSR_ENABLE_TRACE"   # csrc/Makefile's COMMON + the trace switch
for f in api views binning adapter latent_epilogue ply preprocess_backward render_forward render_backward; do /opt/rocm/bin/hipcc $COMMON -c $f.hip -o $OUT/$f.o & done
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -c preprocess.hip -o $OUT/preprocess.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -c sh.hip -o $OUT/sh.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/liblsr_trace.so $OUT/*.o
echo built build_variants/liblsr_trace.so

#!/bin/bash
# Build libaed.so for gfx950 in-tree (the .so travels with the repo snapshot to the GPU box).
set -e
cd "$(dirname "$0")"
ARCH=${AED_ARCH:-gfx950}
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p obj
pids=()
hipcc $FLAGS -c conv_gemm.hip -o obj/conv_gemm.o & pids+=($!)
hipcc $FLAGS -c lin_gemm.hip -o obj/lin_gemm.o & pids+=($!)
hipcc $FLAGS -c conv_gemm_x6.hip -o obj/conv_gemm_x6.o & pids+=($!)
hipcc $FLAGS -c conv_gemm_f8.hip -o obj/conv_gemm_f8.o & pids+=($!)
hipcc $FLAGS -c attention.hip -o obj/attention.o & pids+=($!)
hipcc $FLAGS -c attention_x6.hip -o obj/attention_x6.o & pids+=($!)
hipcc $FLAGS -c norm.hip -o obj/norm.o & pids+=($!)
hipcc $FLAGS -ffp-contract=off -c elementwise.hip -o obj/elementwise.o & pids+=($!)
hipcc $FLAGS -ffp-contract=off -c stable_audio.hip -o obj/stable_audio.o & pids+=($!)
hipcc $FLAGS -c api.hip -o obj/api.o & pids+=($!)
hipcc $FLAGS -c image.hip -o obj/image.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC obj/*.o -o ../libaed.so
echo "built $(cd .. && pwd)/libaed.so"

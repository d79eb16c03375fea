#!/bin/bash
# dev tool (GPU box): for each flag set rebuild gemm_f16.o, relink and run the tile sweep for cfg3
for v in "$@"; do
  (cd 3d-shape-generation_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $v -c gemm_f16.hip -o gemm_f16.o && \
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpcd_hip.so *.o)
  echo "== [$v]"; python tools/bench_gemm.py 3 2>&1 | grep -E "K= 2048|K= 1024 C= 2048|K=  512 C= 1024"
done

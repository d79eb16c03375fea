#!/bin/bash
# dev tool (GPU box): rebuild gemm_f16.o with extra flags, relink, run a tool
cd 3d-shape-generation_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $1 -c gemm_f16.hip -o gemm_f16.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpcd_hip.so *.o
cd ../..; shift; "$@"

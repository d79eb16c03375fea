#!/bin/bash
# dev tool (GPU box): rebuild attention.o with the given full flag sets and time it
cd 3d-shape-generation_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-honor-nans $v -c attention.hip -o attention.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpcd_hip.so core.o gemm_f16.o pointwise.o unet.o metrics.o attention.o latent.o conv3d.o sinkhorn.o skinny.o
  echo "== variant [$v]"; (cd ../..; python tools/dbg_attn.py 2>&1 | grep -E "^1 2048 64"; python tools/bench_attn.py 2>&1 | grep "C=")
done

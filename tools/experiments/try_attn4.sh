#!/bin/bash
# dev tool (GPU box): rebuild attention.o with extra -D flags, run the attention parity tests, then time it
cd 3d-shape-generation_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans $v -c attention.hip -o attention.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpcd_hip.so *.o
  echo "== variant [$v]"; (cd ../..; python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -1; python tools/bench_attn.py 2>&1 | grep "C=256")
done

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o pkadd_next_to_mfma pkadd_next_to_mfma.hip -ldl 2>/dev/null; cd ../..
for bg in skel skel_long; do echo "== aggressor $bg"; timeout -k 10 200 tools/ubench/pkadd_next_to_mfma 10 $bg 2>&1 | cut -c1-175; done

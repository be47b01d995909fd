#!/bin/bash
# builds aggressor variants of the teacher GEMM (FQSS_TDIAG bits, csrc/teacher.hip) and runs the packed-add reproducer next to each
set -o pipefail
cd "$(dirname "$0")/../fqss_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -I../../include -Wno-unused-function"
mkdir -p diag
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL -DFQSS_TDIAG=$v -c teacher.hip -o diag/teacher_v$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag/libfqss_tdiag$v.so $(ls *.o | grep -v '^teacher.o$') diag/teacher_v$v.o
done
ls -la diag/*.so

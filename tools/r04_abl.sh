#!/bin/bash
# timing of library variants (fqss_amd/csrc/variants/libfqss_<name>.so) on the roofline cases matching $PAT (default k_tgemm):
#   PAT=k_tgemm bash tools/r04_abl.sh name ...
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=gpurun_out/r04_abl.txt; : > $out
echo "== product" >> $out
timeout -k 10 120 python3 tools/tgemm_probe.py ${PAT:-k_tgemm} 2>&1 | grep "us$" >> $out || exit 1
for n in "$@"; do
  echo "== $n" >> $out
  FQSS_LIB=$PWD/fqss_amd/csrc/variants/libfqss_$n.so timeout -k 10 120 python3 tools/tgemm_probe.py ${PAT:-k_tgemm} 2>&1 | grep "us$" >> $out || exit 1
done
cat $out

#!/bin/bash
# Does tests/test_gpu_kernels.py::test_grouped_weight_gradients_beside_a_memory_hog catch the round-5 store hazard?  The same test against
# (a) the product library, (b) a variant of csrc/qgemm.hip built with -DFQSS_NO_STORE_NOP (st16_sc1 without its wait states), each with
# wide and with narrow tiles.  Build the variant first (here or on the CPU box):  make -C fqss_amd/csrc variant SRC=qgemm NAME=nonop DEFS=-DFQSS_NO_STORE_NOP
cd "$(dirname "$0")/.."
T=tests/test_gpu_kernels.py::test_grouped_weight_gradients_beside_a_memory_hog
for lib in "" "fqss_amd/csrc/variants/libfqss_nonop.so"; do
  for wide in 1 0; do
    echo "== library ${lib:-product}, FQSS_WGRAD_WIDE=$wide =="
    FQSS_LIB=${lib:+$PWD/$lib} FQSS_WGRAD_WIDE=$wide timeout -k 10 300 python -m pytest "$T" -x -q 2>&1 | grep -E "passed|failed|differ|Error" | head -5
  done
done
exit 0

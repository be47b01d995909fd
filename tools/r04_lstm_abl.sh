#!/bin/bash
# k_lstm_fwd<128> with parts removed (variants/libfqss_labl<n>.so, FQSS_LSTM_ABL bits): what a step's ~3,600 cycles are made of
mkdir -p gpurun_out
out=gpurun_out/r04_lstm_ablation.txt
: > $out
for n in 0 1 2 3 4 8 12 16 32 60 63; do
  if [ $n = 0 ]; then lib=fqss_amd/csrc/libfqss_hip.so; else lib=fqss_amd/csrc/variants/libfqss_labl$n.so; fi
  echo "== ABL $n" >> $out
  FQSS_LIB=$lib timeout -k 10 120 python tools/lstm_probe.py 20 2>&1 | grep "w16 0" >> $out || exit 1
done

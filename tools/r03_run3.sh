#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "[1] reproducer"; tools/ubench/pk_opsel_repro 20 | tee gpurun_out/pk_opsel_repro.log; echo "rc=$?"
echo "[2] gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/gpu_suite.log
echo "[3] bench"; timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-teacher-ahead > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err; python -c "import json;d=json.load(open('gpurun_out/bench_base.json'));print('base ms', d['ms_per_step'])"
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ahead.json 2> gpurun_out/bench_ahead.err; python -c "import json;d=json.load(open('gpurun_out/bench_ahead.json'));print('ahead ms', d['ms_per_step'], d['loss_db'], d['si_sdr_db'])"
echo "[4] diag twin (branchy + SLP) next to the teacher"; FQSS_LIB=$PWD/fqss_amd/csrc/diag/libfqss_diag.so BG=teacher timeout -k 10 200 python tools/diag_streams.py 20 > gpurun_out/diag_teacher.log 2>&1; tail -1 gpurun_out/diag_teacher.log

#!/bin/bash
# round 4: the 256-row teacher GEMM (k_tgemm2) -- parity, then A/B against the round-3 kernel (FQSS_TGEMM_V1=1) in isolation and in the step
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "teacher_gemm" 2>&1 | tail -15 > gpurun_out/r04_tgemm_test.txt; rc=$?
cat gpurun_out/r04_tgemm_test.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_desc_api.py -x -q -k "teacher or desc" 2>&1 | tail -5 || exit 1
echo "--- v2" > gpurun_out/r04_tgemm_probe.txt
timeout -k 10 200 python3 tools/bench_teacher.py 2>&1 | grep "^T1\|^T3" >> gpurun_out/r04_tgemm_probe.txt || exit 1
echo "--- v1 (FQSS_TGEMM_V1=1)" >> gpurun_out/r04_tgemm_probe.txt
FQSS_TGEMM_V1=1 timeout -k 10 200 python3 tools/bench_teacher.py 2>&1 | grep "^T1\|^T3" >> gpurun_out/r04_tgemm_probe.txt || exit 1
cat gpurun_out/r04_tgemm_probe.txt
for v in 0 1 0 1; do
  FQSS_TGEMM_V1=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-workloads --steps 40 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('FQSS_TGEMM_V1=$v ms_per_step',d['ms_per_step'],[ (k['kernel'],k['launch_us']) for k in [d['roofline']]+d['roofline_other_kernels'] if 'tgemm' in k['kernel']])" | tee -a gpurun_out/r04_tgemm_step.txt || exit 1
done

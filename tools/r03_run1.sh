#!/bin/bash
# round-3 GPU session 1: system facts for the two-stream investigation, the look-ahead teacher (test + A/B bench), diagnostic twin
set -o pipefail
mkdir -p gpurun_out
{
  uname -r; cat /sys/module/amdgpu/version 2>/dev/null
  for p in cwsr_enable sched_policy mes hws_max_conc_proc num_kcq debug_evictions; do echo "$p = $(cat /sys/module/amdgpu/parameters/$p 2>/dev/null)"; done
  for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; grep -E "simd_count|lds_size|cwsr|ctl_stack|gfx_target|num_xcc|array_count|cu_per|max_waves|debug_prop|wave_front|num_sdma|unique" $n/properties; done
  /opt/rocm/bin/rocminfo | grep -E "Name:|Compute Unit|LDS|Wavefront|Max Waves|Workgroup Max" | head -40
} > gpurun_out/sysinfo.txt 2>&1
echo "[1] sysinfo done"
timeout -k 10 300 python -m pytest tests/test_gpu_kdstep_path.py -x -q -k "teacher_one_batch_ahead" > gpurun_out/t_ahead.log 2>&1; echo "[2] ahead test rc=$?"; tail -3 gpurun_out/t_ahead.log
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-teacher-ahead > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err; echo "[3a] rc=$?"; python -c "import json;d=json.load(open('gpurun_out/bench_base.json'));print('base ms', d['ms_per_step'])"
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ahead.json 2> gpurun_out/bench_ahead.err; echo "[3b] rc=$?"; python -c "import json;d=json.load(open('gpurun_out/bench_ahead.json'));print('ahead ms', d['ms_per_step'], d['loss_db'], d['si_sdr_db'])"
export FQSS_LIB=$PWD/fqss_amd/csrc/diag/libfqss_diag.so
for bg in teacher tgemm1 tgemm0 tdw mm; do
  BG=$bg timeout -k 10 200 python tools/diag_streams.py 40 > gpurun_out/diag_$bg.log 2>&1; echo "[4] diag $bg rc=$?"; tail -2 gpurun_out/diag_$bg.log
done
timeout -k 10 200 python tools/stress_streams.py mulq_bwd_bias mulq_bwd_prod_out > gpurun_out/stress_branchy.log 2>&1; echo "[5] rc=$?"; tail -3 gpurun_out/stress_branchy.log

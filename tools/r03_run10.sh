set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_long or mha_prep" > gpurun_out/r10_tests.log 2>&1 || { tail -40 gpurun_out/r10_tests.log; exit 1; }
tail -2 gpurun_out/r10_tests.log

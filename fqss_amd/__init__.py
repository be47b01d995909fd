"""fqss_amd -- MI355X-native (gfx950) implementation of the FQSS QAT hot path.

The compute path is hand-written HIP behind the C ABI of include/fqss.h (csrc/libfqss_hip.so);
this package is the Python host that mirrors the reference's `quantization.qat` module API and
`train.py -env asteroid` plugin surface on top of it.  There is NO CPU fallback: every op raises
if the HIP library is missing or a tensor is not on a ROCm device (the CPU restatement lives in
oracle/ and is test infrastructure only).
"""
__version__ = "0.1.0"

"""colibri_amd — host-side Python helpers of the MI355X-native pattern-model builder.

The product is the C-ABI library (include/colibri_hip.h, colibri-core_amd/csrc/) and the C++ face
(colibri-core_amd/host/); this package is the ctypes binding used by tests/bench.py, the synthetic
corpus generators and the multi-GPU exchange driver (torch.distributed over RCCL).
"""

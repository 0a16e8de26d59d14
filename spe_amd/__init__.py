"""spe_amd - MI355X-native (gfx950) implementation of the SPE forward/backward hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd graph, torch.distributed);
all device work on the path goes through the C ABI of ``libspe_hip.so`` (include/spe_hip.h).
There is no CPU fallback: using an op without the built library or off-GPU raises.
"""
__version__ = "0.1.0"

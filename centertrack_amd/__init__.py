"""centertrack_amd: MI355X-native (gfx950) CenterTrack per-frame inference hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all device compute is hand-written HIP behind the C ABI declared in
``include/centertrack_hip.h`` (``centertrack_amd/csrc`` -> ``libcentertrack_hip.so``).
Importing the package does not load the HIP library; the first op call does, and
fails loudly if it is missing (there is no CPU fallback).
"""
__version__ = '0.1.0'

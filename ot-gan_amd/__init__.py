"""otgan_amd -- MI355X-native OT-GAN training hot path.

Host layer (Python, PyTorch-ROCm for device memory / streams / autograd plumbing /
torch.distributed) over the C-ABI library `csrc/libotgan_hip.so` (hand-written gfx950 HIP
kernels, see include/otgan.h).  The sub-modules mirror the reference's own surfaces:

    otgan_amd.utils.matching   <- reference utils/matching.py
    otgan_amd.utils.nn         <- reference utils/nn.py
    otgan_amd.models.dcgan     <- reference models/dcgan.py
    otgan_amd.models.densenet  <- reference models/densenet.py
    otgan_amd.train            <- reference train.py (same command line)

There is no CPU fallback: every operator raises if the HIP library is missing.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"

"""tecogan-pytorch_amd: MI355X-native TecoGAN / FRVSR frame-recurrent hot path.

Hand-written HIP (gfx950) kernels behind a C ABI (include/tecogan_hip.h,
libtecogan_hip.so) with a host-side mirror of the reference's
`codes/models` / `codes/utils` interface for that path.  There is no CPU or
ATen fallback: every op raises if the HIP library is missing.
"""
__version__ = '0.1.0'

import os as _os

# The pipelined clip inference overlaps FNet(t+1) with SRNet(t) on two HIP streams.  With the
# runtime's default STATIC stream -> hardware-queue mapping (GPU_MAX_HW_QUEUES = 4) the second
# stream lands on the first stream's queue on every 4th clip and the overlap is lost for that
# clip (840 -> 716 frames/s; 450 with a high-priority side stream).  With dynamic queue
# assignment the runtime places a newly active stream on an idle queue: steady 840, also next
# to RCCL's streams (measured: tools/pipe_probe.py, tools/dist_probe.py).
# The HIP runtime reads this when it initialises, i.e. at the first GPU call of the process;
# an explicit setting by the user wins.
import sys as _sys

# Was the HIP runtime already up when this package was imported?  Then the variable below
# cannot take effect any more and infer_sequence warns (once) instead of degrading silently.
_torch = _sys.modules.get('torch')
_HIP_UP_BEFORE_IMPORT = bool(_torch is not None and _torch.cuda.is_initialized())
_HAD_SETTING = 'DEBUG_HIP_DYNAMIC_QUEUES' in _os.environ
_os.environ.setdefault('DEBUG_HIP_DYNAMIC_QUEUES', '1')


def dynamic_queues_active():
    """True when the HIP runtime can be expected to run with dynamic stream -> hardware-queue
    assignment: the variable is '1' and it was set before the runtime initialised (by the
    user's environment, or by this import happening before the first GPU call)."""
    if _os.environ.get('DEBUG_HIP_DYNAMIC_QUEUES') != '1':
        return False
    return _HAD_SETTING or not _HIP_UP_BEFORE_IMPORT

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
_os.environ.setdefault('DEBUG_HIP_DYNAMIC_QUEUES', '1')

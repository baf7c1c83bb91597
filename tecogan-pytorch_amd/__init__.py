"""tecogan-pytorch_amd: MI355X-native TecoGAN / FRVSR frame-recurrent hot path.

Hand-written HIP (gfx950) kernels behind a C ABI (include/tecogan_hip.h,
libtecogan_hip.so) with a host-side mirror of the reference's
`codes/models` / `codes/utils` interface for that path.  There is no CPU or
ATen fallback: every op raises if the HIP library is missing.
"""
__version__ = '0.1.0'

# (Rounds 1-2 set DEBUG_HIP_DYNAMIC_QUEUES=1 here.  The problem it papered over was a side stream
# re-created on every clip; since round 3 there is ONE long-lived side stream per device
# (models/networks/tecogan_nets.py side_stream) and importing this package
# no longer touches the environment.)

"""tecogan-pytorch_amd: MI355X-native TecoGAN / FRVSR frame-recurrent hot path.

Hand-written HIP (gfx950) kernels behind a C ABI (include/tecogan_hip.h,
libtecogan_hip.so) with a host-side mirror of the reference's
`codes/models` / `codes/utils` interface for that path.  There is no CPU or
ATen fallback: every op raises if the HIP library is missing.
"""
__version__ = '0.1.0'

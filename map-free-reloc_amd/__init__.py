"""map-free-reloc_amd -- MI355X (gfx950) native drop-in for the feature-matching +
scale-from-depth relative-pose hot path of nianticlabs/map-free-reloc.

Host side is Python (as the reference is); every hot op is a hand-written HIP kernel in
csrc/ reached through the C-ABI declared in include/mfr_hip.h (ctypes, see _lib.py).
There is NO CPU fallback: if libmfr_hip.so is missing or no GPU is visible the solver /
matcher classes raise instead of silently computing elsewhere.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401

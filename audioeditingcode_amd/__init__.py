"""MI355X-native DDPM-inversion audio editing (drop-in for the reference's models.py wrapper API).

Host code is thin Python; all arithmetic of the hot path runs in libaed.so (hand-written HIP for
gfx950, C ABI in include/aed.h).  Importing the package does not require a GPU; running anything does.
"""
__version__ = "0.1.0"

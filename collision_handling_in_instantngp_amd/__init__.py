"""MI355X-native multi-resolution hash-grid encoder with learned (GNGF) collision handling.

Hot path only: hand-written gfx950 HIP kernels behind a C-ABI (include/gngf.h), exposed through
module classes with the reference's constructor / forward signatures.  No CPU fallback."""
__version__ = "0.1.0"

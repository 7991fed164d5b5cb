"""ultragrid_amd -- MI355X-native kernels for UltraGrid's pixel-format-conversion +
block-compression hot path.  The product is libug_mi355x.so (C ABI: include/ug_mi355x.h)
plus the UltraGrid module shims under ultragrid_amd/module/; this Python package is the
test / bench driver's binding of that ABI."""
from . import lib  # noqa: F401

__all__ = ["lib"]

"""ptmi — host-side Python mirror of the reference's compute-pass interface.

The product path is the HIP library (lib/libptmi.so, include/ptmi.h); this
package only binds it (ctypes) and prepares its input blobs. It contains no CPU
rendering fallback: without the HIP library and a GPU every render call raises.
"""
from . import layout  # noqa: F401

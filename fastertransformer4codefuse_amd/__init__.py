"""MI355X (gfx950) native GPT-NeoX / CodeFuse decoder engine -- drop-in for FasterTransformer4CodeFuse's
`libth_gptneox.GptNeoXOp` / `libth_common` hot path.

The compute path is the HIP library `lib/libftcf.so` (C ABI: include/ftcf.h).  There is no CPU fallback: importing
the package works anywhere, but every op raises if the library or a GPU is missing.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]

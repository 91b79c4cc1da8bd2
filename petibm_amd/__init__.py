"""petibm_amd -- MI355X-native linear-solve backend for PetIBM.

The product is the C-ABI shared library ``petibm_amd/lib/libpetibm_amd.so``
(hand-written HIP for gfx950 + RCCL; sources in ``petibm_amd/csrc``, contract in
``include/petibm_amd.h``).  This package holds only the host-side mirror of the
reference's plugin interface (`linsolver`) over that C ABI, and the build
recipe (`build`).  There is NO CPU fallback: without the HIP library, or
without a GPU, every entry point raises.
"""
from .build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]

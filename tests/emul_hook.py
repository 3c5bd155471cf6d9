"""TEST INFRASTRUCTURE: serve CPU tensors with the HOST EMULATION of the kernel source.

raw2logit_amd has one dispatch target (the gfx950 library) and raises for CPU tensors.  The CPU-only test
suite still wants to run the real kernel source (tiling, halos, borders, reductions) against the oracle, so
it builds tests/_build/libr2l_emul.so (tests/emul/r2l_emul.cpp: the same headers compiled by g++ with
R2L_EMUL, every phase looped over tid) and patches ``raw2logit_amd._lib.library_for`` from out here.  CUDA
tensors keep going to the device library; nothing in the package knows about this file."""
import ctypes

from raw2logit_amd import _lib

_ORIGINAL = _lib.library_for
_EMUL = None


class EmulationLibrary(_lib.Library):
    def _check_build(self):
        if self.is_device:
            raise _lib.R2LError(f'{self.path} is a device build, not the host emulation')


def enable(path):
    """path of libr2l_emul.so (or None to switch the patch off) -> the emulation library"""
    global _EMUL
    _EMUL = EmulationLibrary(path) if path else None

    def library_for(t):
        if not t.is_cuda and _EMUL is not None:
            return _EMUL, ctypes.c_void_p(0)
        return _ORIGINAL(t)
    _lib.library_for = library_for if _EMUL is not None else _ORIGINAL
    return _EMUL


def active():
    return _EMUL

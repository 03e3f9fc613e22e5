"""MI355X-native DVAE + GRBM training path.

Layout:
    csrc/        HIP kernels (gfx950) + the C-ABI library ``libdvg.so`` (include/dvg.h)
    _lib.py      ctypes binding of the C-ABI (fails loudly when the library is missing)
    graphs.py    QPU-topology generators, sub-graph selection, colouring, Gibbs plan
    ...          host-side mirror of the reference interface (added module by module)
"""
__version__ = "0.1.0"

"""femo_amd -- MI355X-native engine for femo's PDE-residual hot path.

Layout: ``csrc/`` HIP kernels + C-ABI (``include/femo_hip.h``), ``engine`` the
handle layer over ctypes, ``fea`` / ``csdl_opt`` the mirrors of the reference's
``femo.fea`` / ``femo.csdl_opt`` operator surface.
"""
__version__ = "0.1.0"

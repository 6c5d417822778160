"""Mirror of the reference's ``femo.fea`` package on the HIP engine.

``femo.fea.fea_dolfinx``  -> ``femo_amd.fea.fea_hip``   (class FEA)
``femo.fea.utils_dolfinx`` -> ``femo_amd.fea.utils_hip`` (assemble*, update, solve*, ...)
"""

"""nmma_amd -- MI355X-native batched evaluation of NMMA's EM light-curve likelihood.

Public surface (mirrors ``nmma.em``):

    from nmma_amd.em.model import SVDLightCurveModel
    from nmma_amd.em.em_likelihood import EMTransientLikelihood, OpticalLightCurve
    from nmma_amd.em.systematics import FilterSystematicsHandler

The arithmetic runs in hand-written HIP kernels (``csrc/``) behind a C ABI
(``include/nmma_hip.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"

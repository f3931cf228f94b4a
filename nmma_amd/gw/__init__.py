"""GW messenger of the joint likelihood: waveform + detector projection + inner products fused on the device
(:class:`GravitationalWaveTransientLikelihood`, :class:`GWEngine`) and the reduction for caller-supplied strain
(:class:`GWStrainLikelihood`); see gw_likelihood.py."""
from .detector import Interferometer, greenwich_mean_sidereal_time, site_geometry  # noqa: F401
from .gw_likelihood import (GravitationalWaveTransient, GravitationalWaveTransientLikelihood, GWEngine,  # noqa: F401
                            GWStrainLikelihood, WaveformGenerator)

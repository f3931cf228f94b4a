"""GW messenger of the joint likelihood: the inner-product reduction (see gw_likelihood.py)."""
from .gw_likelihood import GWStrainLikelihood  # noqa: F401

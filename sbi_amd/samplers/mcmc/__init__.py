from sbi_amd.samplers.mcmc.init_strategy import proposal_init, resample_given_potential_fn, sir_init
from sbi_amd.samplers.mcmc.slice_vectorized import SliceSamplerVectorized

__all__ = ["SliceSamplerVectorized", "proposal_init", "sir_init", "resample_given_potential_fn"]

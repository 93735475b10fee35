from sbi_amd.inference.posteriors.direct_posterior import DirectPosterior  # noqa: F401
from sbi_amd.inference.posteriors.mcmc_posterior import MCMCPosterior  # noqa: F401
from sbi_amd.inference.posteriors.rejection_posterior import RejectionPosterior  # noqa: F401
from sbi_amd.inference.trainers.npe.npe import NPE, NPE_C, SNPE  # noqa: F401
from sbi_amd.inference.trainers.vfpe.fmpe import FMPE, posterior_flow_nn  # noqa: F401

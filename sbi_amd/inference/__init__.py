from sbi_amd.inference.posteriors.direct_posterior import DirectPosterior  # noqa: F401
from sbi_amd.inference.trainers.npe.npe import NPE, NPE_C, SNPE  # noqa: F401

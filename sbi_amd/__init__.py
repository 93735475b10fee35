"""sbi_amd -- MI355X-native NSF / NPE hot path behind sbi's own API surface.

Scope: the Neural Spline Flow density estimator's
``log_prob`` / ``sample`` / ``loss`` and the ``NPE.train()`` inner loop, as
hand-written HIP kernels for gfx950 behind ``posterior_nn`` /
``ConditionalDensityEstimator`` / ``NPE`` / ``DirectPosterior``.
"""

__version__ = "0.1.0"

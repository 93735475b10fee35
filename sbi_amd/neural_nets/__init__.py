from sbi_amd.neural_nets.factory import posterior_flow_nn, posterior_nn  # noqa: F401
from sbi_amd.neural_nets.net_builders.estimator_configs import MAFRQSConfig, NSFConfig, ZukoNSFConfig  # noqa: F401

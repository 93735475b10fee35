from sbi_amd.samplers.ode_solvers.dopri5 import odeint_dopri5  # noqa: F401

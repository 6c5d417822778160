"""The form builders of examples/nonlinear_poisson_opt/run_nonlinear_poisson_opt.py (lines
82-142) on the closed catalogue: same names and signatures, so the run script only swaps imports."""
from .forms import L2TrackingFunctional, NonlinearPoissonResidual

ALPHA_1 = 6E-7   # run_nonlinear_poisson_opt.py:80
ALPHA_2 = 2E-6   # :81 (L1 regularisation, unused by the active outputForm)


def interiorResidual(u, v, f):
    """:88-96"""
    return NonlinearPoissonResidual(u, f)


def pdeRes(u, v, f, u_exact=None, weak_bc=False, sym=False, overPenalize=False, beta_value=1e1):
    """:119-126 -> interiorResidual + boundaryResidual(sym, beta_value=1e1)"""
    return NonlinearPoissonResidual(u, f, u_exact=u_exact, weak_bc=weak_bc, sym=sym,
                                    beta_value=beta_value, overPenalize=overPenalize)


def outputForm(u, f, u_exact, alpha=ALPHA_1):
    """:140-142 (L2 regularisation)"""
    return L2TrackingFunctional(u, f, u_exact, alpha)

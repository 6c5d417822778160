"""Round-4 probe (SciPy on the oracle): relative weights of the three parts of the shell's additive preconditioner with the
Hermite-type lattice spaces -- point blocks, node blocks of the levels above the coarse one, exact coarse solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import shell_oracle as so
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_oracle_shell import _roof_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pts, conn, V, K, F, fixed = _roof_problem(n)
M = so.LatticePreconditioner(V, K, fixed, hermite=True)
print("levels", M.levels, "coarse", M.c, flush=True)

def run(wp, wl, wc, per_level=None):
    def apply(r):
        r = r * M.mask
        z = wp * np.einsum("pij,pj->pi", M.Bpt, r.reshape(-1, 3)).ravel() * M.mask
        if M.c >= 0:
            z += wc * (M.P[M.c] @ (M.Ac_inv @ (M.P[M.c].T @ r)))
        for l, B in M.Bl.items():
            g = (M.P[l].T @ r).reshape(-1, 6)
            w = wl if per_level is None else per_level[l]
            z += w * (M.P[l] @ np.einsum("nij,nj->ni", B, g).ravel())
        return z
    old = M.apply
    M.apply = apply
    x, its = M.pcg(F)
    M.apply = old
    return its

print("base", run(1, 1, 1), flush=True)
for wp in (0.5, 0.7, 1.5, 2.0):
    print("point", wp, run(wp, 1, 1), flush=True)
for wl in (0.5, 0.7, 1.5, 2.0):
    print("levels", wl, run(1, wl, 1), flush=True)
for wc in (0.5, 2.0, 4.0):
    print("coarse", wc, run(1, 1, wc), flush=True)
ls = sorted(M.Bl)
for combo in ([1, 1, 0.5], [0.5, 1, 1], [1, 0.5, 1], [2, 1, 1], [1, 1, 2], [0, 1, 1], [1, 0, 1], [0, 0, 1]):
    if len(combo) >= len(ls):
        pl = {l: combo[len(combo) - len(ls) + i] for i, l in enumerate(ls)}
        print("per level", pl, run(1, 1, 1, pl), flush=True)

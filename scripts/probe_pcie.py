"""Host <-> device rates of Vec.set / Vec.get (pageable NumPy arrays, the CSDL boundary)."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from femo_amd.engine import Context, Vec

ctx = Context(0)
for n in (10_077_696, 59_630_250):
    a = np.random.default_rng(0).random(n)
    v = Vec(ctx, n)
    v.set(a); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        v.set(a)
    ctx.sync()
    h2d = 3 * n * 8 / (time.perf_counter() - t0) / 1e9
    out = v.get()
    t0 = time.perf_counter()
    for _ in range(3):
        out = v.get()
    d2h_fresh = 3 * n * 8 / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(3):
        b = np.empty(n)
        b[:] = 0.0
    touch = 3 * n * 8 / (time.perf_counter() - t0) / 1e9
    print(f"n={n}: H2D {h2d:.1f} GB/s, D2H into a fresh array {d2h_fresh:.1f} GB/s, first touch of a fresh array {touch:.1f} GB/s", flush=True)

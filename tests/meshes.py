"""Unstructured test meshes (test infrastructure)."""
import numpy as np


def l_shape_mesh(m: int = 12, grade: float = 1.6, jitter: float = 0.25, seed: int = 3):
    """L-shaped domain [-1,1]^2 minus (0,1]x(0,1], triangulated from a (2m)^2 grid with alternating diagonals,
    graded towards the re-entrant corner (x -> sign(x)|x|^grade), interior vertices jittered, vertices AND cells
    randomly renumbered.  Returns x, conn, boundary edges (vertex pairs) and their tags: 1 = outer boundary,
    2 = the two edges meeting at the re-entrant corner."""
    rng = np.random.default_rng(seed)
    n1 = 2 * m + 1
    g = np.linspace(-1.0, 1.0, n1)
    X, Y = np.meshgrid(g, g, indexing="xy")
    ii, jj = np.meshgrid(np.arange(2 * m), np.arange(2 * m), indexing="xy")
    keep = ~((ii >= m) & (jj >= m))
    ii, jj = ii[keep], jj[keep]
    v00 = jj * n1 + ii
    v10, v01, v11 = v00 + 1, v00 + n1, v00 + n1 + 1
    flip = ((ii + jj) % 2 == 0)
    t1 = np.where(flip[:, None], np.stack([v00, v10, v11], 1), np.stack([v00, v10, v01], 1))
    t2 = np.where(flip[:, None], np.stack([v00, v11, v01], 1), np.stack([v10, v11, v01], 1))
    conn = np.concatenate([t1, t2])
    used = np.unique(conn)
    remap = -np.ones(n1 * n1, dtype=np.int64)
    remap[used] = np.arange(used.size)
    conn = remap[conn]
    x = np.stack([X.ravel()[used], Y.ravel()[used]], 1)
    # boundary edges before moving anything
    e = np.concatenate([conn[:, [0, 1]], conn[:, [1, 2]], conn[:, [2, 0]]])
    es = np.sort(e, axis=1)
    uniq, cnt = np.unique(es, axis=0, return_counts=True)
    bedges = uniq[cnt == 1]
    mid = x[bedges].mean(axis=1)
    reentrant = ((np.abs(mid[:, 0]) < 1e-12) & (mid[:, 1] > 0)) | ((np.abs(mid[:, 1]) < 1e-12) & (mid[:, 0] > 0))
    tags = np.where(reentrant, 2, 1).astype(np.int32)
    bverts = np.zeros(x.shape[0], bool)
    bverts[np.unique(bedges)] = True
    h = 1.0 / m
    xj = x.copy()
    xj[~bverts] += rng.uniform(-1, 1, size=(np.count_nonzero(~bverts), 2)) * jitter * h
    xg = np.sign(xj) * np.abs(xj) ** grade
    # orientation stays positive? (alternating diagonals + jitter < h/2 keep the cells valid) -- renumber
    pv = rng.permutation(x.shape[0])
    x2 = np.empty_like(xg)
    x2[pv] = xg
    conn2 = pv[conn][rng.permutation(conn.shape[0])]
    return x2, conn2.astype(np.int32), pv[bedges].astype(np.int32), tags

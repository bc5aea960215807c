"""Synthetic triangle soups that are hard on a BVH builder (tests of the builders rt_create picks by itself for big scenes):
coincident triangles, collinear centroids with sizes six orders of magnitude apart, a dense cluster beside a few scene-sized
triangles, a flat sheet (one axis without extent), a uniform random soup.  Materials, lights and camera are the `cubes`
preset's."""
import numpy as np

import pyrt

KINDS = ("identical", "line", "corner", "flat", "random")


def soup(kind, n, seed=1):
    rng = np.random.default_rng(seed)
    base = pyrt.Scene("cubes", 16, 16).arrays()
    if kind == "identical":
        # n copies of ONE triangle (the same three vertices): no split separates anything
        pos = np.array([[0.1, 0.2, -3.0], [1.3, 0.1, -3.2], [0.4, 1.5, -2.9]], np.float32)
        tri = np.tile(np.array([0, 1, 2], np.uint32), (n, 1))
    else:
        if kind == "line":
            c = np.zeros((n, 3))
            c[:, 0] = np.linspace(-50, 50, n)
            size = 10.0 ** rng.uniform(-4, 2, n)
        elif kind == "corner":
            c = rng.uniform(-1e-3, 1e-3, (n, 3)) + [5, 5, -5]
            size = np.full(n, 2e-4)
            c[:10] = rng.uniform(-1, 1, (10, 3))
            size[:10] = 100.0
        elif kind == "flat":
            g = int(np.ceil(np.sqrt(n)))
            ix, iy = np.divmod(np.arange(n), g)
            c = np.stack([ix * 0.01, iy * 0.01, np.full(n, -2.0)], 1)
            size = np.full(n, 0.02)
        elif kind == "random":
            c = rng.uniform(-10, 10, (n, 3))
            size = 10.0 ** rng.uniform(-3, 0.5, n)
        else:
            raise ValueError(kind)
        off = rng.uniform(-1, 1, (n, 3, 3)) * size[:, None, None]
        if kind == "flat":
            off[:, :, 2] = 0
        pos = (c[:, None, :] + off).reshape(-1, 3).astype(np.float32)
        tri = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(pos), 1))
    return pyrt.ArrayScene(pos, nrm, tri, [0, n], [0, len(pos)], base["materials"][:1], base["lights"], base["camera"])


def soup_rays(scene, n, seed=2):
    """Rays aimed at the soup: origins around it, directions through random vertices (most hit something), a few at random."""
    rng = np.random.default_rng(seed)
    a = scene.arrays()
    pos = a["pos"]
    lo, hi = pos.min(0), pos.max(0)
    ext = np.maximum(hi - lo, 0.25 * (hi - lo).max() + 1e-3)  # (a flat sheet is looked at from beside it, not from within)
    rays = np.zeros(n, pyrt.RAY_DTYPE)
    o = rng.uniform(lo - ext, hi + ext, (n, 3))
    tri = a["tri"][rng.integers(0, len(a["tri"]), n)]
    w = rng.dirichlet([1, 1, 1], n)
    target = (pos[tri] * w[:, :, None]).sum(1)  # a point INSIDE a random triangle
    d = target - o
    rnd = rng.random(n) < 0.1
    d[rnd] = rng.normal(size=(int(rnd.sum()), 3))
    rays["origin"], rays["direction"] = o.astype(np.float32), d.astype(np.float32)
    return rays

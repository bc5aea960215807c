"""Shared ray-batch generator of the parity tests (GPU and CPU side)."""
import numpy as np

import pyrt


def ray_batch(scene, n, seed):
    """Camera rays, rays leaving surface points exactly (no epsilon), random rays,
    axis-parallel rays (zero direction components) and a NaN ray."""
    rng = np.random.default_rng(seed)
    a = scene.arrays()
    rays = np.zeros(n, pyrt.RAY_DTYPE)
    cam = a["camera"]
    u, v = rng.random(n, np.float32), rng.random(n, np.float32)
    d = cam[1] + u[:, None] * cam[2] + v[:, None] * cam[3] - cam[0]
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays["origin"], rays["direction"] = cam[0], d.astype(np.float32)
    # surface starts: barycentric points of random triangles
    m = n // 2
    t = rng.integers(0, len(a["tri"]), m)
    b = rng.random((m, 2), np.float32)
    b[b.sum(1) > 1] = 1 - b[b.sum(1) > 1]
    P = a["pos"][a["tri"][t]]
    w = (1 - b[:, 0] - b[:, 1]).astype(np.float32)
    pts = w[:, None] * P[:, 0] + b[:, 0:1] * P[:, 1] + b[:, 1:2] * P[:, 2]
    rays["origin"][:m] = pts.astype(np.float32)
    dirs = rng.normal(size=(m, 3)).astype(np.float32)
    dirs[::3] /= np.linalg.norm(dirs[::3], axis=1, keepdims=True)
    dirs[1::5] = (np.array([0.0, -0.3, 1.1], np.float32) - pts[1::5]).astype(np.float32)  # towards light 2
    rays["direction"][:m] = dirs
    q = n // 16
    rays["direction"][m:m + q] = rng.choice(np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 1, 1], [-1, 0, 1]], np.float32), q)
    rays["origin"][m:m + q] = rng.uniform(-1.4, 1.4, (q, 3)).astype(np.float32)
    rays["direction"][m + q] = np.nan
    return rays

"""Deterministic synthetic clouds for the configurations whose data cannot travel to the GPU box
(SURVEY.md 8d): a closed, non-symmetric star-shaped surface r(theta,phi) = 0.6 + 0.35*tanh(sum a_k Y_k),
bounded in [-1,1]^3 like the reference's normalised scans.

  S1 "bunny-scale":  seed 20241223, M = N = 40 000, V = 300
  S2 "1M":           seed 20241224, M = N = 1 000 000, V = 512
The source is an independent resample of the same surface, moved by a known rigid motion
(axis-angle (0.9,-1.7,2.3) rad is applied as  target = R * source + t, i.e. the engine must recover
R, t) plus N(0, noise^2) per coordinate.
"""
import numpy as np

S1 = dict(seed=20241223, M=40000, N=40000, V=300)
S2 = dict(seed=20241224, M=1000000, N=1000000, V=512)
GT_AXIS_ANGLE = np.array([0.9, -1.7, 2.3])
GT_T = np.array([0.21, -0.15, 0.07])


def _rodrigues(v):
    th = np.linalg.norm(v)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _surface(rng, n, coef, amp=0.35):
    # uniform directions; radius from low-order harmonics with odd terms (x, y, z, xyz) so that the
    # shape has no rotational or point symmetry (a near-sphere would make Go-ICP's early exit
    # accept any rotation).  Radius stays in [0.25, 0.95].  Not area-uniform; irrelevant here.
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d.T
    basis = np.stack([x, y, z, x * y, y * z, 3 * z * z - 1, x * x - y * y, 5 * x * y * z], axis=1)
    r = 0.6 + amp * np.tanh(basis @ coef)
    return d * r[:, None]


def make_pair(seed=S1["seed"], M=S1["M"], N=S1["N"], noise=0.002, V=None, amp=0.35):
    """-> (target (M,3) f32, source (N,3) f32, R_gt (3,3), t_gt (3,)) with target ~= R_gt*source + t_gt.
    amp = relief of the surface (0.35: strongly non-symmetric, ICP's basin of convergence is wide; smaller values
    approach a sphere: many shallow local minima, the BnB has to dig for the true basin)."""
    rng = np.random.default_rng(seed)
    coef = rng.uniform(-1, 1, size=8)
    target = _surface(rng, M, coef, amp)
    moved = _surface(rng, N, coef, amp) + rng.normal(scale=noise, size=(N, 3))
    R = _rodrigues(GT_AXIS_ANGLE)
    # source = R^T (moved - t)  =>  R*source + t = moved, which lies on the target surface
    source = (moved - GT_T) @ R
    return target.astype(np.float32), source.astype(np.float32), R, GT_T.copy()

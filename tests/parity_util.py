"""Tolerances and comparison helpers shared by the GPU parity tests (fp32 device arithmetic vs the fp64 reference;
BASELINE.md section 4)."""
import numpy as np

POS_TOL = 1e-4      # positions
DIST_TOL = 2e-4     # observation distance
GUARD = 1e-3        # guard band around z = 0 and |delta| = tol inside which discrete outputs are not compared


def assert_obs_close(obs, ref_obs, elbow, points, alive):
    """obs, ref_obs (N,3K); elbow (N,3), points (N,K,3) and alive (N,K) from the oracle (pre-pickup state)."""
    n, k = alive.shape
    o = obs.reshape(n, k, 3).astype(np.float64)
    r = ref_obs.reshape(n, k, 3)
    m_ = np.abs(elbow[:, None, :] - points)
    rho_xy = np.hypot(m_[..., 0], m_[..., 1])
    dist = r[..., 0]
    with np.errstate(divide="ignore"):
        tol_r = np.maximum(1e-3, 57.3 * 1e-4 / rho_xy)
        tol_th = np.maximum(1e-3, 57.3 * 1e-4 / dist)
    dead = ~alive
    assert np.all(o[dead] == 0.0)
    assert np.all(np.abs(o[..., 0] - r[..., 0])[alive] <= DIST_TOL), np.abs(o[..., 0] - r[..., 0])[alive].max()
    err_r = np.abs(o[..., 1] - r[..., 1])
    err_t = np.abs(o[..., 2] - r[..., 2])
    assert np.all((err_r <= tol_r)[alive]), (err_r - tol_r)[alive].max()
    assert np.all((err_t <= tol_th)[alive]), (err_t - tol_th)[alive].max()

"""Trajectory I/O and ATE / RTE evaluation.

The reference writes `results/txts/output_<name>_offset<o>.txt` with lines `t px py pz qx qy qz qw`
(`msckf.py:152-160`) and publishes ATE/RTE numbers in `results/metrics_summary.csv`, but its metric
script is not in the repository (git-ignored).  This module defines the metric explicitly:
ATE = RMSE of position after a least-squares SE(3) alignment (Umeyama without scale) of the estimated
positions to the ground truth at associated timestamps; RTE = RMSE of the translation part of the
relative-pose error over a fixed frame delta, both trajectories expressed in their own frames."""
import numpy as np


def format_state_line(timestamp, position, orientation):
    """The reference's trajectory line (msckf.py:153-158)."""
    return '%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n' % (timestamp, position[0], position[1], position[2],
                                                          orientation[0], orientation[1], orientation[2], orientation[3])


def load_trajectory_txt(path):
    """float64[n, 8] (t px py pz qx qy qz qw).  The reference opens its file in append mode
    (msckf.py:159), so a file may hold several concatenated runs: only the last run (after the last
    backwards time jump) is returned."""
    rows = []
    with open(path) as f:
        for line in f:
            v = line.split()
            if len(v) == 8:
                rows.append([float(x) for x in v])
    a = np.array(rows, dtype=np.float64).reshape(-1, 8)
    if len(a) > 1:
        back = np.nonzero(np.diff(a[:, 0]) < 0)[0]
        if len(back):
            a = a[back[-1] + 1:]
    return a


def associate(t_est, t_ref, max_dt=0.01):
    """Nearest-timestamp association; returns index arrays (i_est, i_ref)."""
    t_ref = np.asarray(t_ref)
    j = np.searchsorted(t_ref, t_est)
    j = np.clip(j, 1, len(t_ref) - 1)
    left = np.abs(t_ref[j - 1] - t_est) <= np.abs(t_ref[j] - t_est)
    j = np.where(left, j - 1, j)
    ok = np.abs(t_ref[j] - t_est) <= max_dt
    return np.nonzero(ok)[0], j[ok]


def umeyama_se3(src, dst):
    """R, t minimising sum |R src_i + t - dst_i|^2 (no scale)."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    H = (src - mu_s).T @ (dst - mu_d)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mu_d - R @ mu_s


def ate(est, ref, max_dt=0.01):
    """est, ref: float[n, >=4] with columns t, px, py, pz.  Returns dict(rmse, mean, std, n, R, t)."""
    ie, ir = associate(est[:, 0], ref[:, 0], max_dt)
    if len(ie) < 3:
        raise ValueError('fewer than 3 associated poses')
    P, Q = est[ie, 1:4], ref[ir, 1:4]
    R, t = umeyama_se3(P, Q)
    e = np.linalg.norm((P @ R.T + t) - Q, axis=1)
    return dict(rmse=float(np.sqrt((e ** 2).mean())), mean=float(e.mean()), std=float(e.std()), n=int(len(e)), R=R, t=t)


def rte(est, ref, delta=10, max_dt=0.01):
    """Relative translation error over `delta` associated frames."""
    ie, ir = associate(est[:, 0], ref[:, 0], max_dt)
    P, Q = est[ie, 1:4], ref[ir, 1:4]
    if len(P) <= delta:
        raise ValueError('trajectory shorter than delta')
    R, _ = umeyama_se3(P, Q)
    dP = (P[delta:] - P[:-delta]) @ R.T
    dQ = Q[delta:] - Q[:-delta]
    e = np.linalg.norm(dP - dQ, axis=1)
    return dict(rmse=float(np.sqrt((e ** 2).mean())), mean=float(e.mean()), std=float(e.std()), n=int(len(e)))

"""Generates tests/golden/msckf_*.npz by IMPORTING THE REFERENCE FILTER (src/msckf.py, src/feature/,
src/utils.py are importable with numpy + scipy only) and recording its outputs.

    python tests/golden/make_msckf_golden.py            # needs /root/reference (build container only)

Inputs are regenerated deterministically by the tests from uav_airvision_amd.synth (seeded), so only
the reference's OUTPUTS (plus small explicit inputs for the unit vectors) are stored.  The reference
source never leaves /root/reference; nothing here copies it.
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = '/root/reference/src'

from uav_airvision_amd.config import ConfigEuRoC                         # noqa: E402
from uav_airvision_amd.synth import SyntheticFeatureStream, replay_features   # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    import msckf as ref_msckf          # noqa: F401  (creates nothing at import)
    import utils as ref_utils
    import feature as ref_feature
    return ref_msckf, ref_utils, ref_feature


def utils_vectors(ref_utils, out):
    rng = np.random.default_rng(42)
    q = rng.normal(size=(24, 4)); q[:4] *= 3.0                      # un-normalised inputs on purpose
    q2 = rng.normal(size=(24, 4))
    dth = rng.normal(scale=0.2, size=(24, 3)); dth[:3] *= 20.0         # both branches of small_angle_quaternion
    v0 = rng.normal(size=(24, 3)); v1 = rng.normal(size=(24, 3))
    v1[0] = -v0[0]; v1[1] = v0[1] * 2.0; v0[2] = np.array([0, 1.0, 0]); v1[2] = np.array([0, -1.0, 0])
    out['u_q'] = q; out['u_q2'] = q2; out['u_dth'] = dth; out['u_v0'] = v0; out['u_v1'] = v1
    out['u_R'] = np.array([ref_utils.to_rotation(x) for x in q])
    out['u_q_of_R'] = np.array([ref_utils.to_quaternion(R) for R in out['u_R']])
    # rotations that visit every branch of to_quaternion
    Rs = []
    for ax in np.eye(3):
        for ang in (0.3, 2.9):
            Rs.append(ref_utils.to_rotation(np.array([*(ax * np.sin(ang / 2)), np.cos(ang / 2)])))
    out['u_Rb'] = np.array(Rs)
    out['u_q_of_Rb'] = np.array([ref_utils.to_quaternion(R) for R in Rs])
    out['u_qmul'] = np.array([ref_utils.quaternion_multiplication(a, b) for a, b in zip(q, q2)])
    out['u_small'] = np.array([ref_utils.small_angle_quaternion(d) for d in dth])
    out['u_two'] = np.array([ref_utils.from_two_vectors(a, b) for a, b in zip(v0, v1)])


class _Cam(object):
    pass


def triangulation_vectors(ref_feature, ref_msckf, cfg, out):
    """Feature.initialize_position on explicit inputs (stored)."""
    fs = SyntheticFeatureStream(cfg, seed=11, n_frames=30, n_features=40, pixel_sigma=0.8)
    ref_feature.BaseFeature.R_cam0_cam1 = cfg.T_cn_cnm1[:3, :3]
    ref_feature.BaseFeature.t_cam0_cam1 = cfg.T_cn_cnm1[:3, 3]
    import utils as ref_utils
    cams = {}
    for k in range(30):
        R_i_w, p = fs.truth(k)
        c = _Cam()
        c.orientation = ref_utils.to_quaternion((R_i_w @ fs.T_c0_i[:3, :3]).T)
        c.position = p + R_i_w @ fs.T_c0_i[:3, 3]
        cams[k] = c
    obs = {}
    for k in range(30):
        for f in fs.frame(k).features:
            obs.setdefault(f.id, {})[k] = np.array([f.u0, f.v0, f.u1, f.v1])
    rng = np.random.default_rng(5)
    ids = [i for i, o in obs.items() if len(o) >= 2][:60]
    cases_obs, cases_ids, res_pos, res_ok = [], [], [], []
    for n, fid in enumerate(ids):
        feat = ref_feature.Feature(fid, cfg.optimization_config)
        o = dict(obs[fid])
        if n % 7 == 3:                                  # corrupt a few: far outliers / behind-camera cases
            k0 = next(iter(o))
            o[k0] = o[k0] + rng.normal(0, 0.3, 4)
        feat.observations = o
        ok = feat.initialize_position(cams)
        arr = np.full((30, 4), np.nan)
        for k, z in o.items():
            arr[k] = z
        cases_obs.append(arr); cases_ids.append(fid); res_pos.append(feat.position.copy()); res_ok.append(bool(ok))
    out['t_cam_q'] = np.array([cams[k].orientation for k in range(30)])
    out['t_cam_p'] = np.array([cams[k].position for k in range(30)])
    out['t_obs'] = np.array(cases_obs)
    out['t_pos'] = np.array(res_pos)
    out['t_ok'] = np.array(res_ok)


class _Thr(np.float64):
    """A chi-square threshold that notes what it is compared with.  gating_test (msckf.py:604-612) evaluates
    `gamma < table[dof]` with gamma an np.float64; Python gives the reflected comparison of a SUBCLASS operand
    precedence, so this __gt__ sees the reference's own gamma without touching its code or its result."""
    sink = None

    def __gt__(self, other):
        res = float(other) < float(self)
        if _Thr.sink is not None:
            _Thr.sink.append((float(other), bool(res)))
        return res


def record_calls(ref_msckf, flt):
    """Per-call outputs of the reference's gating_test / measurement_update / process_model:
    gamma and the gate decision of every gated feature (msckf.py:604-612), delta_x, stacked shape and P+ of every
    update (:548-602), P after every IMU sample (:275-339).  delta_x is a local of measurement_update; it is
    recovered as the owner array of the slice handed to small_angle_quaternion at :576 (delta_x[:21][:3].base)."""
    calls = dict(gamma=[], gate_ok=[], gate_dof=[], gate_frame=[],
                 upd_frame=[], upd_m=[], upd_n=[], upd_dx=[], upd_Pdiag=[], upd_Ptrace=[], upd_P=[],
                 pm_frame=[], pm_P11diag=[], pm_Ptrace=[], pm_P12fro=[], pm_P11=[])
    cur = dict(frame=0, dx=None, in_update=False, n_pm=0)
    sink = []
    _Thr.sink = sink
    flt.chi_squared_test_table = {k: _Thr(v) for k, v in flt.chi_squared_test_table.items()}
    orig_gate, orig_upd, orig_pm = flt.gating_test, flt.measurement_update, flt.process_model
    orig_saq = ref_msckf.small_angle_quaternion

    def saq(dtheta):
        if cur['in_update'] and cur['dx'] is None:
            base = dtheta.base
            assert base is not None and base.ndim == 1 and len(base) >= 21
            cur['dx'] = np.array(base)
        return orig_saq(dtheta)
    ref_msckf.small_angle_quaternion = saq

    def gate(H, r, dof):
        n0 = len(sink)
        ok = orig_gate(H, r, dof)
        assert len(sink) == n0 + 1 and sink[-1][1] == bool(ok)
        calls['gamma'].append(sink[-1][0]); calls['gate_ok'].append(bool(ok)); calls['gate_dof'].append(dof)
        calls['gate_frame'].append(cur['frame'])
        return ok

    def upd(H, r):
        if len(H) == 0 or len(r) == 0:
            return orig_upd(H, r)
        cur['in_update'], cur['dx'] = True, None
        orig_upd(H, r)
        cur['in_update'] = False
        P = flt.state_server.state_cov
        dx = np.zeros(21 + 6 * 30); dx[:len(cur['dx'])] = cur['dx']
        pd = np.zeros(21 + 6 * 30); pd[:len(P)] = np.diag(P)
        calls['upd_frame'].append(cur['frame']); calls['upd_m'].append(H.shape[0]); calls['upd_n'].append(H.shape[1])
        calls['upd_dx'].append(dx); calls['upd_Pdiag'].append(pd); calls['upd_Ptrace'].append(np.trace(P))
        if len(calls['upd_frame']) % 16 == 1:
            calls['upd_P'].append((len(calls['upd_frame']) - 1, P.copy()))

    def pm(time, m_gyro, m_acc):
        orig_pm(time, m_gyro, m_acc)
        P = flt.state_server.state_cov
        calls['pm_frame'].append(cur['frame']); calls['pm_P11diag'].append(np.diag(P)[:21].copy())
        calls['pm_Ptrace'].append(np.trace(P)); calls['pm_P12fro'].append(np.linalg.norm(P[:21, 21:]))
        if cur['n_pm'] % 50 == 0:
            calls['pm_P11'].append((cur['n_pm'], P[:21, :21].copy()))
        cur['n_pm'] += 1
    flt.gating_test, flt.measurement_update, flt.process_model = gate, upd, pm

    def finish(out):
        ref_msckf.small_angle_quaternion = orig_saq
        _Thr.sink = None
        for k in ('gamma', 'gate_ok', 'gate_dof', 'gate_frame', 'upd_frame', 'upd_m', 'upd_n', 'upd_dx', 'upd_Pdiag', 'upd_Ptrace',
                  'pm_frame', 'pm_P11diag', 'pm_Ptrace', 'pm_P12fro'):
            out['c_' + k] = np.array(calls[k])
        for i, P in calls['upd_P']:
            out['c_updP_%d' % i] = P
        for i, P in calls['pm_P11']:
            out['c_pmP11_%d' % i] = P
    return cur, finish


def run_filter(ref_msckf, cfg, seed, n_frames, n_features, full_P_at=(), calls=False, **kw):
    fs = SyntheticFeatureStream(cfg, seed=seed, n_frames=n_frames, n_features=n_features, **kw)
    ref_msckf.IMUState.next_id = 0
    flt = ref_msckf.MSCKF(cfg)
    cur, finish = record_calls(ref_msckf, flt) if calls else (None, None)
    rec = {k: [] for k in ('t', 'q', 'p', 'v', 'bg', 'ba', 'R_ic', 't_ci', 'Pdiag', 'Ptrace', 'ncam', 'nmap', 'published')}
    fullP = {}
    frame = [0]

    def on(msg):
        if cur is not None:
            cur['frame'] = frame[0]
        res = flt.feature_callback(msg)
        s = flt.state_server.imu_state
        P = flt.state_server.state_cov
        rec['t'].append(msg.timestamp); rec['q'].append(np.array(s.orientation)); rec['p'].append(np.array(s.position))
        rec['v'].append(np.array(s.velocity)); rec['bg'].append(np.array(s.gyro_bias)); rec['ba'].append(np.array(s.acc_bias))
        rec['R_ic'].append(np.array(s.R_imu_cam0)); rec['t_ci'].append(np.array(s.t_cam0_imu))
        rec['Pdiag'].append(np.diag(P)[:21].copy()); rec['Ptrace'].append(np.trace(P)); rec['ncam'].append(len(flt.state_server.cam_states))
        rec['nmap'].append(len(flt.map_server)); rec['published'].append(res is not None)
        if frame[0] in full_P_at:
            fullP[frame[0]] = P.copy()
        frame[0] += 1

    replay_features(fs, [flt.imu_callback], on)
    out = {k: np.array(v) for k, v in rec.items()}
    for k, P in fullP.items():
        out['P_%d' % k] = P
    out['gravity'] = np.array(ref_msckf.IMUState.gravity)
    if finish is not None:
        finish(out)
    return out


def main():
    only_new = '--all' not in sys.argv              # the round-1 files are left alone unless --all is given
    cfg = ConfigEuRoC()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                               # the reference ctor creates results/txts in the cwd
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref_msckf, ref_utils, ref_feature = import_reference()
                if not only_new:
                    u = {}
                    utils_vectors(ref_utils, u)
                    triangulation_vectors(ref_feature, ref_msckf, cfg, u)
                    e1 = run_filter(ref_msckf, cfg, seed=0, n_frames=150, n_features=100, full_P_at=(25, 60, 149))
                    e2 = run_filter(ref_msckf, cfg, seed=3, n_frames=45, n_features=300, full_P_at=(44,))
                # round 3: per-call vectors (gamma, delta_x, P+, process_model P) along a run that prunes, and the
                # configs[4] shape: 1,500 features per frame (the > 1500 rows cut msckf.py:667-668, the uncapped
                # camera-pruning update :712-786)
                c1 = run_filter(ref_msckf, cfg, seed=5, n_frames=60, n_features=100, full_P_at=(59,), calls=True, outlier_rate=0.02)
                c2 = run_filter(ref_msckf, cfg, seed=9, n_frames=30, n_features=1500, full_P_at=(29,), calls=True, outlier_rate=0.005)
        finally:
            os.chdir(cwd)
    gdir = os.path.join(ROOT, 'tests', 'golden')
    if not only_new:
        np.savez_compressed(os.path.join(gdir, 'msckf_units.npz'), **u)
        np.savez_compressed(os.path.join(gdir, 'msckf_e2e_seed0_n100.npz'), seed=0, n_frames=150, n_features=100, **e1)
        np.savez_compressed(os.path.join(gdir, 'msckf_e2e_seed3_n300.npz'), seed=3, n_frames=45, n_features=300, **e2)
        print('units:', {k: v.shape for k, v in u.items()})
        print('e2e n100: published', int(e1['published'].sum()), 'ncam max', int(e1['ncam'].max()), 'P trace last', float(e1['Ptrace'][-1]))
        print('e2e n300: published', int(e2['published'].sum()), 'ncam max', int(e2['ncam'].max()))
    c2 = {k: v for k, v in c2.items() if not k.startswith('c_pm')}          # the IMU part is the same code at any feature count
    np.savez_compressed(os.path.join(gdir, 'msckf_calls_seed5_n100.npz'), seed=5, n_frames=60, n_features=100, outlier_rate=0.02, **c1)
    np.savez_compressed(os.path.join(gdir, 'msckf_calls_seed9_n1500.npz'), seed=9, n_frames=30, n_features=1500, outlier_rate=0.005, **c2)
    for name, c in (('calls n100', c1), ('calls n1500', c2)):
        print(name, 'gates', len(c['c_gamma']), 'passed', int(c['c_gate_ok'].sum()), 'updates', len(c['c_upd_m']),
              'rows max', int(c['c_upd_m'].max()), 'cols max', int(c['c_upd_n'].max()), 'ncam max', int(c['ncam'].max()))


if __name__ == '__main__':
    main()

"""Shared by the CPU and GPU filter tests: run a filter object (numpy oracle or drop-in MSCKF) over the seeded feature
stream of a tests/golden/msckf_calls_*.npz fixture and compare its per-call outputs with the REFERENCE's
(recorded by tests/golden/make_msckf_golden.py:record_calls from /root/reference/src/msckf.py:275-339, 548-612)."""
import numpy as np

from uav_airvision_amd.synth import SyntheticFeatureStream, replay_features


def stream_of(cfg, g):
    return SyntheticFeatureStream(cfg, seed=int(g['seed']), n_frames=int(g['n_frames']), n_features=int(g['n_features']),
                                  outlier_rate=float(g['outlier_rate']))


def run_recording(flt, fs, get_P, get_state):
    """flt needs .debug with 'gamma' (list, appended per gate) and either a `measurement_update` method that sets
    debug['delta_x'] (oracle.msckf_np.OracleMSCKF, tests/stepwise_msckf.MSCKF: wrapped here to catch the covariance right after
    every update) or `capture_debug` (the drop-in view over the batched filter: the library hands out delta_x and P+ of the
    frame's updates in debug['updates'])."""
    rec = dict(gamma=[], gate_frame=[], upd_frame=[], upd_dx=[], upd_Pdiag=[], upd_Ptrace=[], upd_P={}, frames=[])
    k = [0]

    def note(dx, P):
        rec['upd_frame'].append(k[0]); rec['upd_dx'].append(np.array(dx))
        rec['upd_Pdiag'].append(np.diag(P).copy()); rec['upd_Ptrace'].append(np.trace(P))
        rec['upd_P'][len(rec['upd_frame']) - 1] = P.copy() if (len(rec['upd_frame']) - 1) % 16 == 0 else None

    view = hasattr(flt, 'capture_debug')
    if view:
        flt.capture_debug(True)
    else:
        orig_upd = flt.measurement_update

        def upd(H, r, *a, **kw):
            flt.debug.pop('delta_x', None)
            out = orig_upd(H, r, *a, **kw)
            if 'delta_x' in flt.debug:
                note(flt.debug['delta_x'], get_P())
            return out
        flt.measurement_update = upd

    def on(msg):
        flt.debug['gamma'] = []
        res = flt.feature_callback(msg)
        rec['gamma'].extend(flt.debug['gamma']); rec['gate_frame'].extend([k[0]] * len(flt.debug['gamma']))
        if view:
            for _rows, dx, P in flt.debug.get('updates', []):
                note(dx, P)
        st = get_state(); st['published'] = res is not None
        rec['frames'].append(st)
        k[0] += 1
    replay_features(fs, [flt.imu_callback], on)
    return rec


def compare_calls(rec, g, tol_gamma, tol_dx, tol_P, tol_state):
    """gamma / delta_x / P+ are basis-invariant: they must agree whatever null-space basis or
    factorisation the implementation uses."""
    assert len(rec['gamma']) == len(g['c_gamma'])
    assert np.array_equal(rec['gate_frame'], g['c_gate_frame'])
    gam = np.array(rec['gamma'])
    assert np.allclose(gam, g['c_gamma'], rtol=tol_gamma, atol=1e-12), np.abs(gam / g['c_gamma'] - 1).max()
    assert (~g['c_gate_ok']).sum() > 5                                   # the fixture holds rejected features
    assert np.array_equal(rec['upd_frame'], g['c_upd_frame'])
    for i, dx in enumerate(rec['upd_dx']):
        n = int(g['c_upd_n'][i])
        assert len(dx) == n, (i, len(dx), n)
        ref = g['c_upd_dx'][i][:n]
        assert np.abs(dx - ref).max() <= tol_dx * max(np.abs(ref).max(), 1e-3), (i, np.abs(dx - ref).max(), np.abs(ref).max())
        assert np.allclose(rec['upd_Pdiag'][i], g['c_upd_Pdiag'][i][:n], rtol=tol_P, atol=1e-16), i
        assert abs(rec['upd_Ptrace'][i] / g['c_upd_Ptrace'][i] - 1) < tol_P
        if 'c_updP_%d' % i in g.files and rec['upd_P'].get(i) is not None:
            Pr = g['c_updP_%d' % i]
            assert np.abs(rec['upd_P'][i] - Pr).max() <= tol_P * np.abs(Pr).max(), i
    fr = rec['frames']
    assert np.array_equal([f['ncam'] for f in fr], g['ncam']) and np.array_equal([f['nmap'] for f in fr], g['nmap'])
    assert np.array_equal([f['published'] for f in fr], g['published'])
    for key in ('q', 'p', 'v', 'bg', 'ba', 'R_ic', 't_ci'):
        err = np.abs(np.array([f[key] for f in fr]) - g[key]).max()
        assert err < tol_state, (key, err)

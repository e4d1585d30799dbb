"""Parity of the HIP MSCKF back-end (through the C ABI and the drop-in classes) against the CPU
oracle and against golden vectors produced by the reference filter itself.

Tolerance (fp64 everywhere on both sides; the GPU uses Householder-QR null spaces and Cholesky
solves where the reference uses SVD and LU, and tree-ordered sums where numpy is sequential):
 * single operators: 1e-9 relative,
 * end-to-end after 150 frames (hundreds of chained updates): 1e-6 on pose / velocity, 1e-6
   relative on the covariance -- five orders tighter than the fp64->fp32 bar of the north star."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module', params=['view', 'stepwise'])
def dropin(request):
    """(filter module, feature module).  'view' = the product: dropin/msckf.py, a view over the batched C++/HIP filter with one
    stream; 'stepwise' = tests/stepwise_msckf.py, the Python driver that sequences the single-filter operator API
    (av_msckf_propagate / _feature_blocks / _update ...), so those entry points stay under the same reference vectors."""
    import torch
    assert torch.cuda.is_available()
    d = os.path.join(ROOT, 'uav_airvision_amd', 'dropin')
    if d not in sys.path:
        sys.path.insert(0, d)
    import msckf as dmsckf
    import feature as dfeature
    if request.param == 'stepwise':
        import stepwise_msckf
        return stepwise_msckf, dfeature
    return dmsckf, dfeature


class _Cam(object):
    pass


def test_triangulation_matches_reference_vectors(dropin, cfg):
    dmsckf, dfeature = dropin
    u = np.load(os.path.join(G, 'msckf_units.npz'))
    cams = {}
    for k in range(len(u['t_cam_q'])):
        c = _Cam(); c.orientation = u['t_cam_q'][k]; c.position = u['t_cam_p'][k]
        cams[k] = c
    dfeature.BaseFeature.R_cam0_cam1 = cfg.T_cn_cnm1[:3, :3]
    dfeature.BaseFeature.t_cam0_cam1 = cfg.T_cn_cnm1[:3, 3]
    for i in range(len(u['t_obs'])):
        f = dfeature.Feature(i, cfg.optimization_config)
        for k in range(u['t_obs'].shape[1]):
            if not np.isnan(u['t_obs'][i, k, 0]):
                f.observations[k] = u['t_obs'][i, k]
        ok = f.initialize_position(cams)
        assert ok == bool(u['t_ok'][i]), i
        assert np.allclose(f.position, u['t_pos'][i], rtol=1e-8, atol=1e-9), (i, f.position, u['t_pos'][i])


def _drive(flt, fs, per_frame):
    from uav_airvision_amd.synth import replay_features
    replay_features(fs, [flt.imu_callback], lambda m: per_frame(flt.feature_callback(m)))


def _state(flt_state, P):
    s = flt_state
    return dict(q=s.orientation.copy(), p=s.position.copy(), v=s.velocity.copy(), bg=s.gyro_bias.copy(), ba=s.acc_bias.copy(),
                R_ic=s.R_imu_cam0.copy(), t_ci=s.t_cam0_imu.copy(), P=P)


def test_operator_parity_along_a_run(dropin, cfg):
    """Drive the GPU filter and the numpy oracle in lock-step, copying the oracle's state into the
    GPU filter before every frame, so each frame tests the operators on identical inputs."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    dmsckf, _ = dropin
    fs = SyntheticFeatureStream(cfg, seed=7, n_frames=40, n_features=80)
    gpu = dmsckf.MSCKF(cfg, write_trajectory=False)
    if hasattr(gpu, 'capture_debug'):
        gpu.capture_debug(True)
    ora = OracleMSCKF(cfg)
    it = iter(fs.imu); pend = next(it, None)
    worst = 0.0
    n_updates = 0
    for k in range(fs.n_frames):
        msg = fs.frame(k)
        while pend is not None and pend.timestamp <= msg.timestamp:
            gpu.imu_callback(pend); ora.imu_callback(pend)
            pend = next(it, None)
        ora.debug.pop('delta_x', None); gpu.debug.pop('delta_x', None)
        ora.debug['gamma'] = []; gpu.debug['gamma'] = []
        ra = ora.feature_callback(msg)
        rg = gpu.feature_callback(msg)
        assert (ra is None) == (rg is None)
        Pg, Po = gpu.state_server.state_cov, ora.state_cov
        assert Pg.shape == Po.shape
        scale = np.abs(Po).max()
        errP = np.abs(Pg - Po).max() / scale
        sg, so = gpu.state_server.imu_state, ora.imu_state
        errx = max(np.abs(sg.position - so.position).max(), np.abs(sg.velocity - so.velocity).max(),
                   np.abs(sg.orientation - so.orientation).max())
        worst = max(worst, errP, errx)
        assert errP < 1e-7 and errx < 1e-8, (k, errP, errx)
        assert len(gpu.map_server) == len(ora.map_server) and list(gpu.state_server.cam_states) == list(ora.cam_states)
        if 'delta_x' in ora.debug:
            n_updates += 1
            assert np.allclose(gpu.debug['delta_x'], ora.debug['delta_x'], rtol=1e-6, atol=1e-10)
        if ora.debug['gamma']:
            assert np.allclose(gpu.debug['gamma'], ora.debug['gamma'], rtol=1e-7)
    assert n_updates > 10
    gpu.close()


@pytest.mark.parametrize('name', ['msckf_e2e_seed0_n100.npz', 'msckf_e2e_seed3_n300.npz'])
def test_end_to_end_matches_reference_golden(dropin, cfg, name):
    from uav_airvision_amd.synth import SyntheticFeatureStream
    dmsckf, _ = dropin
    g = np.load(os.path.join(G, name))
    fs = SyntheticFeatureStream(cfg, seed=int(g['seed']), n_frames=int(g['n_frames']), n_features=int(g['n_features']))
    flt = dmsckf.MSCKF(cfg, write_trajectory=False)
    rec = []
    _drive(flt, fs, lambda res: rec.append((_state(flt.state_server.imu_state, None), len(flt.state_server.cam_states), len(flt.map_server), res is not None)))
    assert np.array_equal([r[1] for r in rec], g['ncam'])
    assert np.array_equal([r[2] for r in rec], g['nmap'])
    assert np.array_equal([r[3] for r in rec], g['published'])
    for key, tol in (('q', 1e-6), ('p', 1e-6), ('v', 1e-6), ('bg', 1e-7), ('ba', 1e-6), ('R_ic', 1e-6), ('t_ci', 1e-6)):
        err = np.abs(np.array([r[0][key] for r in rec]) - g[key]).max()
        assert err < tol, (key, err)
    P = flt.state_server.state_cov
    last = 'P_%d' % (int(g['n_frames']) - 1)
    assert np.abs(P - g[last]).max() <= 1e-6 * np.abs(g[last]).max()
    flt.close()


@pytest.mark.parametrize('groups,store', [(1, 'device'), (2, 'device'), (1, 'device-w64'), (1, 'host'), (2, 'host')])
def test_batched_filters_match_reference_golden_and_oracle(cfg, groups, store, monkeypatch):
    """Three independent streams stepped together by the C++/HIP batched filter: stream 0 reproduces the
    reference's own golden run; all streams follow the numpy oracle frame by frame.  groups=2 splits the batch
    into two concurrently stepped stream groups (streams {0,1} and {2}), as large batches are by default.
    store='host' runs the same checks through the host-bookkeeping path (AV_MSCKF_STORE=host: the observation map and the
    selections on the host, the numeric phases by the same kernels) -- the fallback for shapes the device-resident store
    does not cover; it stays in the library only as long as it stays tested.  store='device-w64': the per-stream kernels of the
    device-resident path as ONE wavefront per stream (AV_DK_WG=64), the launch shape batches of 1,024 streams and more take by
    themselves -- three streams would get four wavefronts."""
    monkeypatch.setenv('AV_MSCKF_GROUPS', str(groups))
    monkeypatch.delenv('AV_DK_WG', raising=False)
    if store == 'device-w64':
        monkeypatch.setenv('AV_DK_WG', '64')
        store = 'device'
    if store == 'host':
        monkeypatch.setenv('AV_MSCKF_STORE', 'host')
    else:
        monkeypatch.delenv('AV_MSCKF_STORE', raising=False)
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    g = np.load(os.path.join(G, 'msckf_e2e_seed0_n100.npz'))
    n_frames = 150
    streams = [SyntheticFeatureStream(cfg, seed=0, n_frames=n_frames, n_features=100),
               SyntheticFeatureStream(cfg, seed=21, n_frames=n_frames, n_features=60, motion_scale=1.5),
               SyntheticFeatureStream(cfg, seed=22, n_frames=n_frames, n_features=140)]
    S = len(streams)
    bat = BatchedMSCKF(cfg, S)
    assert bat.device_resident() == (store == 'device')
    oras = [OracleMSCKF(cfg) for _ in streams]
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    cap = 256
    worst = 0.0
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                oras[i].imu_callback(pend[i])
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        out = bat.step(ids, uv, nf, [m.timestamp for m in msgs])
        for i, m in enumerate(msgs):
            r = oras[i].feature_callback(m)
            assert (r is not None) == bool(out[i, 0])
            s = oras[i].imu_state
            n, ncam, nmap = bat.sizes(i)
            assert ncam == len(oras[i].cam_states) and nmap == len(oras[i].map_server) and n == oras[i].state_cov.shape[0], (k, i)
            err = max(np.abs(out[i, 2:5] - s.position).max(), np.abs(out[i, 5:9] - s.orientation).max(), np.abs(out[i, 9:12] - s.velocity).max())
            worst = max(worst, err)
            assert err < 1e-6, (k, i, err)
            # every target of the state injection (msckf.py:576-595): biases, extrinsics, the camera states of the window
            gs = bat.get_state(i)
            assert np.abs(gs['bg'] - s.gyro_bias).max() < 1e-7 and np.abs(gs['ba'] - s.acc_bias).max() < 1e-6, (k, i)
            assert np.abs(gs['R_ic'] - s.R_imu_cam0).max() < 1e-6 and np.abs(gs['t_ci'] - s.t_cam0_imu).max() < 1e-6, (k, i)
            assert list(gs['cam_ids']) == list(oras[i].cam_states.keys()), (k, i)
            if len(gs['cam_ids']):
                assert np.abs(gs['cam_q'] - np.array([c.orientation for c in oras[i].cam_states.values()])).max() < 1e-6
                assert np.abs(gs['cam_p'] - np.array([c.position for c in oras[i].cam_states.values()])).max() < 1e-6
            if i == 0 and out[0, 0]:       # stream 0 against the REFERENCE's own run (golden)
                assert np.abs(gs['bg'] - g['bg'][k]).max() < 1e-7 and np.abs(gs['ba'] - g['ba'][k]).max() < 1e-6, k
                assert np.abs(gs['R_ic'] - g['R_ic'][k]).max() < 1e-6 and np.abs(gs['t_ci'] - g['t_ci'][k]).max() < 1e-6, k
        # out[., 1] is imu_state.timestamp (msckf.py:268: the last IMU sample integrated, <= one IMU period before the frame)
        assert 0.0 <= g['t'][k] - out[0, 1] < 0.0051
        assert np.abs(out[0, 2:5] - g['p'][k]).max() < 1e-6 and np.abs(out[0, 5:9] - g['q'][k]).max() < 1e-6
    for i in range(S):
        P, Po = bat.get_cov(i), oras[i].state_cov
        assert np.abs(P - Po).max() <= 1e-6 * np.abs(Po).max()
    assert np.abs(bat.get_cov(0) - g['P_149']).max() <= 1e-6 * np.abs(g['P_149']).max()
    bat.close()


def test_batched_filter_inactive_stream_and_online_reset():
    """Stream 1 gets its IMU late (no gravity initialisation for the first frames -> feature_callback returns
    None, msckf.py:182-183); stream 0 runs with a tiny position_std_threshold so online_reset fires
    (msckf.py:821-843).  Both must follow the numpy oracle."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.config import ConfigEuRoC
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    cfg = ConfigEuRoC()
    cfg.position_std_threshold = 0.012
    n_frames = 60
    streams = [SyntheticFeatureStream(cfg, seed=31, n_frames=n_frames, n_features=80),
               SyntheticFeatureStream(cfg, seed=32, n_frames=n_frames, n_features=80)]
    bat = BatchedMSCKF(cfg, 2)
    oras = [OracleMSCKF(cfg), OracleMSCKF(cfg)]
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    held = []                      # stream 1's IMU is withheld for the first 8 frames
    resets = 0
    inactive_frames = 0
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                if i == 1 and k < 8:
                    held.append(pend[i])
                else:
                    for q in (held if i == 1 else []):
                        oras[1].imu_callback(q); si.append(1); ts.append(q.timestamp); gy.append(q.angular_velocity); ac.append(q.linear_acceleration)
                    if i == 1:
                        held = []
                    oras[i].imu_callback(pend[i]); si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((2, 512), np.int64); uv = np.zeros((2, 512, 4)); nf = np.zeros(2, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        out = bat.step(ids, uv, nf, [m.timestamp for m in msgs])
        for i, m in enumerate(msgs):
            ncam_before = len(oras[i].cam_states)
            r = oras[i].feature_callback(m)
            assert (r is not None) == bool(out[i, 0]), (k, i)
            if r is None:
                inactive_frames += 1
                continue
            if len(oras[i].cam_states) == 0 and ncam_before > 0:
                resets += 1
            n, ncam, nmap = bat.sizes(i)
            assert (n, ncam, nmap) == (oras[i].state_cov.shape[0], len(oras[i].cam_states), len(oras[i].map_server)), (k, i)
            s = oras[i].imu_state
            assert np.abs(out[i, 2:5] - s.position).max() < 1e-6 and np.abs(out[i, 5:9] - s.orientation).max() < 1e-6, (k, i)
    assert inactive_frames >= 5 and resets >= 1
    for i in range(2):
        assert np.abs(bat.get_cov(i) - oras[i].state_cov).max() <= 1e-6 * max(1e-3, np.abs(oras[i].state_cov).max())
    bat.close()


def test_batched_filter_1500_features_per_frame(cfg):
    """BASELINE configs[4] shape: 1500 features per frame.  The two-camera prune then stacks ~7000 rows per stream
    (beyond the register-resident QR shapes: the global-memory fallback runs) and the lost-feature path hits the
    `> 1500 rows` cut of msckf.py:667-668.  Compared with the numpy oracle frame by frame."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    n_frames = 24
    st = SyntheticFeatureStream(cfg, seed=41, n_frames=n_frames, n_features=1500)
    bat = BatchedMSCKF(cfg, 1, rows_cap=8192)
    ora = OracleMSCKF(cfg)
    it = iter(st.imu); pend = next(it, None)
    cap = 1536
    pruned = False
    for k in range(n_frames):
        m = st.frame(k)
        si, ts, gy, ac = [], [], [], []
        while pend is not None and pend.timestamp <= m.timestamp:
            ora.imu_callback(pend)
            si.append(0); ts.append(pend.timestamp); gy.append(pend.angular_velocity); ac.append(pend.linear_acceleration)
            pend = next(it, None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((1, cap), np.int64); uv = np.zeros((1, cap, 4)); nf = np.array([len(m.features)], np.int32)
        for j, f in enumerate(m.features):
            ids[0, j] = f.id; uv[0, j] = (f.u0, f.v0, f.u1, f.v1)
        ncam_before = len(ora.cam_states)
        out = bat.step(ids, uv, nf, [m.timestamp])
        r = ora.feature_callback(m)
        assert (r is not None) == bool(out[0, 0])
        if r is None:
            continue
        pruned = pruned or len(ora.cam_states) < ncam_before
        n, ncam, nmap = bat.sizes(0)
        assert (n, ncam, nmap) == (ora.state_cov.shape[0], len(ora.cam_states), len(ora.map_server)), k
        s = ora.imu_state
        err = max(np.abs(out[0, 2:5] - s.position).max(), np.abs(out[0, 5:9] - s.orientation).max(), np.abs(out[0, 9:12] - s.velocity).max())
        assert err < 1e-6, (k, err)
    assert pruned, 'the run must reach the camera-state pruning path'
    P, Po = bat.get_cov(0), ora.state_cov
    assert np.abs(P - Po).max() <= 1e-6 * np.abs(Po).max()
    bat.close()


def test_queued_steps_run_ahead_and_match_the_oracle(cfg, monkeypatch):
    """av_msckf_batch_submit / _wait: the two stream groups consume their queues independently (up to one frame apart);
    every retired step must equal the oracle's, exactly as with the blocking step."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    monkeypatch.setenv('AV_MSCKF_GROUPS', '2')
    n_frames = 45
    streams = [SyntheticFeatureStream(cfg, seed=51 + i, n_frames=n_frames, n_features=60 + 30 * i) for i in range(4)]
    S = len(streams)
    bat = BatchedMSCKF(cfg, S)
    oras = [OracleMSCKF(cfg) for _ in streams]
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    cap = 192
    outs, refs = [], []
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                oras[i].imu_callback(pend[i])
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        outs.append(bat.submit(ids, uv, nf, [m.timestamp for m in msgs]))
        bat.wait(1)
        row = []
        for i, m in enumerate(msgs):
            r = oras[i].feature_callback(m)
            s = oras[i].imu_state
            row.append((r is not None, s.position.copy(), s.orientation.copy(), s.velocity.copy()))
        refs.append(row)
    bat.wait(0)
    for k in range(n_frames):
        for i in range(S):
            pub, p, q, v = refs[k][i]
            assert bool(outs[k][i, 0]) == pub, (k, i)
            if pub:
                err = max(np.abs(outs[k][i, 2:5] - p).max(), np.abs(outs[k][i, 5:9] - q).max(), np.abs(outs[k][i, 9:12] - v).max())
                assert err < 1e-6, (k, i, err)
    for i in range(S):
        P, Po = bat.get_cov(i), oras[i].state_cov
        assert np.abs(P - Po).max() <= 1e-6 * np.abs(Po).max()
    bat.close()


def test_queued_step_errors_surface_at_wait(cfg, monkeypatch):
    """A capacity error raised inside a queued step (here: a message capacity whose camera-pruning update cannot fit
    rows_cap) is reported by the wait that retires it, with the library's message, and the batch can still be closed."""
    from uav_airvision_amd._native import AirvisionError
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    monkeypatch.setenv('AV_MSCKF_GROUPS', '2')
    streams = [SyntheticFeatureStream(cfg, seed=71 + i, n_frames=30, n_features=100) for i in range(2)]
    bat = BatchedMSCKF(cfg, 2, rows_cap=1664)
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    raised = None
    for k in range(30):
        msgs = [s.frame(k) for s in streams]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((2, 512), np.int64); uv = np.zeros((2, 512, 4)); nf = np.zeros(2, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        bat.submit(ids, uv, nf, [m.timestamp for m in msgs])
        try:
            bat.wait(1)
        except AirvisionError as e:
            raised = e
            break
    assert raised is not None and 'rows_cap' in str(raised)
    bat.close()


def test_batched_long_run_with_feature_table_compaction(cfg):
    """300 frames with a high feature turnover: the insertion-ordered feature slab of the batched filter compacts and its
    id table rehashes several times; map size, camera states and the state must keep following the numpy oracle."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    n_frames = 300
    streams = [SyntheticFeatureStream(cfg, seed=81, n_frames=n_frames, n_features=50, motion_scale=2.0),
               SyntheticFeatureStream(cfg, seed=82, n_frames=n_frames, n_features=90, motion_scale=1.5)]
    S = len(streams)
    bat = BatchedMSCKF(cfg, S)
    oras = [OracleMSCKF(cfg) for _ in streams]
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    cap = 128
    seen = [set(), set()]
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                oras[i].imu_callback(pend[i])
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1); seen[i].add(f.id)
        out = bat.step(ids, uv, nf, [m.timestamp for m in msgs])
        for i, m in enumerate(msgs):
            r = oras[i].feature_callback(m)
            assert (r is not None) == bool(out[i, 0])
            if r is None:
                continue
            n, ncam, nmap = bat.sizes(i)
            assert (n, ncam, nmap) == (oras[i].state_cov.shape[0], len(oras[i].cam_states), len(oras[i].map_server)), (k, i)
            s = oras[i].imu_state
            err = max(np.abs(out[i, 2:5] - s.position).max(), np.abs(out[i, 5:9] - s.orientation).max(), np.abs(out[i, 9:12] - s.velocity).max())
            assert err < 2e-6, (k, i, err)
    # turnover check: many more features came and went than are alive at the end (so tombstones had to be compacted)
    for i in range(S):
        assert len(seen[i]) > 6 * max(1, len(oras[i].map_server)), (len(seen[i]), len(oras[i].map_server))
        P, Po = bat.get_cov(i), oras[i].state_cov
        assert np.abs(P - Po).max() <= 2e-6 * np.abs(Po).max()
    bat.close()


def test_overflow_pool_exhaustion_stops_the_stream_with_a_capacity_error(cfg, monkeypatch):
    """The overflow pool is a capacity like the camera slots and the feature table: a stream that cannot get its rows stops with
    AV_E_CAPACITY, loudly, and the other streams of the batch go on.  (AV_MSCKF_POOL_ROWS shrinks the pool for this test.)"""
    from uav_airvision_amd import _native as N
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream, feature_msg_t
    monkeypatch.setenv('AV_MSCKF_POOL_ROWS', '64')
    n_frames = 24
    streams = [SyntheticFeatureStream(cfg, seed=91, n_frames=n_frames, n_features=150),
               SyntheticFeatureStream(cfg, seed=92, n_frames=n_frames, n_features=40)]
    S = len(streams)
    bat = BatchedMSCKF(cfg, S, rows_cap=2048)
    if not bat.device_resident():
        bat.close(); pytest.skip('host-bookkeeping path (AV_MSCKF_STORE=host): no pool')
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    cap = 192
    failed_at = None
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        if k == 17:
            msgs[0] = feature_msg_t(msgs[0].timestamp, [])            # stream 0 loses ~130 long tracks at once: ~5 k rows
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        out = bat.step(ids, uv, nf, [m.timestamp for m in msgs])
        if out[0, 0] < 0 and failed_at is None:
            failed_at = k
        if k > 2:
            assert out[1, 0] == 1.0, k                                 # stream 1 keeps publishing
    assert failed_at == 17, failed_at
    code, msg = bat.stream_status(0)
    assert code == N.AV_E_CAPACITY and 'pool' in msg, (code, msg)
    assert bat.stream_status(1) == (0, '')
    bat.close()


def test_blank_frame_drops_every_track_at_once(cfg):
    """A blank / blurred frame: the feature message is empty, so every live track is lost in the same frame and the
    lost-feature candidates reserve far more rows than rows_cap (here ~3-4 k against 2048).  The reference handles any
    number (msckf.py:614-676: gate in map order, stop after > 1500 stacked rows); the device-resident batch gives such a
    stream rows from the overflow pool all streams share (the host-bookkeeping path gates first and stores only the stacked
    features second).  Must equal the numpy oracle on every frame, before, at and after."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream, feature_msg_t
    n_frames = 48
    streams = [SyntheticFeatureStream(cfg, seed=91, n_frames=n_frames, n_features=150),
               SyntheticFeatureStream(cfg, seed=92, n_frames=n_frames, n_features=60)]
    S = len(streams)
    bat = BatchedMSCKF(cfg, S, rows_cap=2048)
    oras = [OracleMSCKF(cfg) for _ in streams]
    its = [iter(s.imu) for s in streams]
    pend = [next(it, None) for it in its]
    cap = 192
    # before the first camera pruning (frame 19) the tracks are long: stream 0's 130 candidates reserve ~5.5 k rows at frame
    # 17, stream 1's 58 candidates ~2.7 k at frame 18, against rows_cap 2048; a later blank frame (41: tracks shortened by
    # the pruning, no candidate with >= 3 observations) takes the ordinary route
    blank = {0: (17, 41), 1: (18, 41)}
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        msgs = [feature_msg_t(m.timestamp, []) if k in blank[i] else m for i, m in enumerate(msgs)]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                oras[i].imu_callback(pend[i])
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        out = bat.step(ids, uv, nf, [m.timestamp for m in msgs])
        for i, m in enumerate(msgs):
            r = oras[i].feature_callback(m)
            assert (r is not None) == bool(out[i, 0])
            n, ncam, nmap = bat.sizes(i)
            assert (n, ncam, nmap) == (oras[i].state_cov.shape[0], len(oras[i].cam_states), len(oras[i].map_server)), (k, i)
            s = oras[i].imu_state
            err = max(np.abs(out[i, 2:5] - s.position).max(), np.abs(out[i, 5:9] - s.orientation).max(), np.abs(out[i, 9:12] - s.velocity).max())
            assert err < 1e-6, (k, i, err)
    c = bat.counters()
    assert c['two_pass_streams'] >= 2, c                 # the blank frames really outgrew rows_cap (pool / two-pass route taken)
    assert c['devbuf_growths'] == 0, c                   # nothing was reallocated after the first step's reserve
    for i in range(S):
        P, Po = bat.get_cov(i), oras[i].state_cov
        assert np.abs(P - Po).max() <= 1e-6 * np.abs(Po).max()
    bat.close()


def test_default_trajectory_file_is_written_and_parses(dropin, cfg, tmp_path, monkeypatch):
    """The default MSCKF(config) writes results/txts/output_<DATASET_NAME>_offset<TIME_OFFSET>.txt in append mode, one
    line per published frame in the reference's format (msckf.py:10-16, 152-160); evaluate.load_trajectory_txt reads it
    back and, after a second run appended to the same file, returns the last run only."""
    dmsckf, _ = dropin
    from uav_airvision_amd import evaluate
    from uav_airvision_amd.synth import SyntheticFeatureStream
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('DATASET_NAME', 'SYN_07')
    monkeypatch.setenv('TIME_OFFSET', '40')
    fs = SyntheticFeatureStream(cfg, seed=5, n_frames=30, n_features=60)
    runs = []
    for _ in range(2):
        flt = dmsckf.MSCKF(cfg)                          # write_trajectory defaults to True
        poses = []
        _drive(flt, fs, lambda r: poses.append(None if r is None else (flt.state_server.imu_state.timestamp, flt.state_server.imu_state.position.copy(),
                                                                       flt.state_server.imu_state.orientation.copy())))
        flt.close()
        runs.append([p for p in poses if p is not None])
    path = tmp_path / 'results' / 'txts' / 'output_SYN_07_offset40.txt'
    assert path.exists()
    raw = [l for l in open(path).read().splitlines() if l.strip()]
    assert len(raw) == len(runs[0]) + len(runs[1]) and len(runs[0]) == 30
    assert all(len(l.split()) == 8 for l in raw)
    tr = evaluate.load_trajectory_txt(str(path))
    assert tr.shape == (30, 8)                           # the concatenated first run is dropped at the backwards time jump
    for row, (t, p, q) in zip(tr, runs[1]):
        assert abs(row[0] - t) < 1e-6 and np.abs(row[1:4] - p).max() < 1e-9 and np.abs(row[4:8] - q).max() < 1e-9


# ---- per-call vectors of the REFERENCE (round 3; tests/golden/msckf_calls_*.npz) -------------------------------------

def _dropin_calls(dmsckf, cfg, g):
    from _calls import run_recording, stream_of
    flt = dmsckf.MSCKF(cfg, write_trajectory=False)

    def state():
        s = flt.state_server.imu_state
        return dict(q=s.orientation.copy(), p=s.position.copy(), v=s.velocity.copy(), bg=s.gyro_bias.copy(), ba=s.acc_bias.copy(),
                    R_ic=s.R_imu_cam0.copy(), t_ci=s.t_cam0_imu.copy(), ncam=len(flt.state_server.cam_states), nmap=len(flt.map_server))
    rec = run_recording(flt, stream_of(cfg, g), lambda: flt.state_server.state_cov, state)
    return rec, flt


@pytest.mark.parametrize('name', ['msckf_calls_seed5_n100.npz', 'msckf_calls_seed9_n1500.npz'])
def test_per_call_gamma_dx_P_match_reference_golden(dropin, cfg, name):
    """HIP feature_kernel / update kernels against the reference's OWN per-call outputs (not the oracle's): gamma of every
    gated feature (msckf.py:604-612), delta_x and P+ of every update (:548-602), incl. rejected outliers, the > 1500-row
    cut (:667-668) and 6,485-row camera-pruning updates (:712-786) in the 1,500-feature fixture."""
    from _calls import compare_calls
    dmsckf, _ = dropin
    g = np.load(os.path.join(G, name))
    rec, flt = _dropin_calls(dmsckf, cfg, g)
    compare_calls(rec, g, tol_gamma=1e-6, tol_dx=1e-6, tol_P=1e-6, tol_state=1e-6)
    last = 'P_%d' % (int(g['n_frames']) - 1)
    P = flt.state_server.state_cov
    assert np.abs(P - g[last]).max() <= 1e-6 * np.abs(g[last]).max()
    flt.close()


def test_batched_filter_matches_reference_golden_at_1500_features(cfg):
    """The batched C++/HIP filter on the reference's own 1,500-features-per-frame run (configs[4] shape):
    pose / velocity / sizes per frame and the final covariance against /root/reference's outputs."""
    from _calls import stream_of
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    g = np.load(os.path.join(G, 'msckf_calls_seed9_n1500.npz'))
    st = stream_of(cfg, g)
    bat = BatchedMSCKF(cfg, 1, rows_cap=8192)
    it = iter(st.imu); pend = next(it, None)
    cap = 1536
    for k in range(st.n_frames):
        m = st.frame(k)
        si, ts, gy, ac = [], [], [], []
        while pend is not None and pend.timestamp <= m.timestamp:
            si.append(0); ts.append(pend.timestamp); gy.append(pend.angular_velocity); ac.append(pend.linear_acceleration)
            pend = next(it, None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((1, cap), np.int64); uv = np.zeros((1, cap, 4)); nf = np.array([len(m.features)], np.int32)
        for j, f in enumerate(m.features):
            ids[0, j] = f.id; uv[0, j] = (f.u0, f.v0, f.u1, f.v1)
        out = bat.step(ids, uv, nf, [m.timestamp])
        assert bool(out[0, 0]) == bool(g['published'][k])
        if not out[0, 0]:
            continue
        n, ncam, nmap = bat.sizes(0)
        assert (ncam, nmap) == (int(g['ncam'][k]), int(g['nmap'][k])), k
        err = max(np.abs(out[0, 2:5] - g['p'][k]).max(), np.abs(out[0, 5:9] - g['q'][k]).max(), np.abs(out[0, 9:12] - g['v'][k]).max())
        assert err < 1e-6, (k, err)
    last = 'P_%d' % (st.n_frames - 1)
    assert np.abs(bat.get_cov(0) - g[last]).max() <= 1e-6 * np.abs(g[last]).max()
    bat.close()


def _feed(bat, oras, streams, its, pend, k, cap):
    """One frame of every stream into the batch (and the IMU samples up to it into batch + oracles)."""
    S = len(streams)
    msgs = [st.frame(k) if k < st.n_frames else None for st in streams]
    si, ts, gy, ac = [], [], [], []
    for i, m in enumerate(msgs):
        while m is not None and pend[i] is not None and pend[i].timestamp <= m.timestamp:
            if oras[i] is not None:
                oras[i].imu_callback(pend[i])
            si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
            pend[i] = next(its[i], None)
    if si:
        bat.push_imu(si, ts, gy, ac)
    ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32); tt = np.full(S, -1.0)
    for i, m in enumerate(msgs):
        if m is None:
            continue
        nf[i] = len(m.features); tt[i] = m.timestamp
        ids[i, :nf[i]] = [f.id for f in m.features]
        uv[i, :nf[i]] = [(f.u0, f.v0, f.u1, f.v1) for f in m.features]
    return msgs, bat.step(ids, uv, nf, tt)


def test_batched_filter_window_of_24_camera_states_and_the_limit():
    """one back-end pass holds 144 columns = 24 camera states.  At max_cam_state_size = 24 long lost-feature
    updates (m > 144, n_c up to 138) must follow the oracle; 25 and more are refused at create, not computed wrongly."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd._native import AirvisionError
    from uav_airvision_amd.config import ConfigEuRoC
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    cfg = ConfigEuRoC()
    cfg.max_cam_state_size = 28
    with pytest.raises(AirvisionError, match='max_cam_states'):
        BatchedMSCKF(cfg, 1)
    cfg.max_cam_state_size = 24
    n_frames = 60
    streams = [SyntheticFeatureStream(cfg, seed=81, n_frames=n_frames, n_features=150), SyntheticFeatureStream(cfg, seed=82, n_frames=n_frames, n_features=60)]
    bat = BatchedMSCKF(cfg, 2)
    oras = [OracleMSCKF(cfg), OracleMSCKF(cfg)]
    its = [iter(s.imu) for s in streams]; pend = [next(it, None) for it in its]
    big = 0
    for k in range(n_frames):
        msgs, out = _feed(bat, oras, streams, its, pend, k, 256)
        for i, m in enumerate(msgs):
            oras[i].debug.pop('delta_x', None)
            r = oras[i].feature_callback(m)
            assert (r is not None) == bool(out[i, 0])
            if r is None:
                continue
            s = oras[i].imu_state
            assert bat.sizes(i) == (oras[i].state_cov.shape[0], len(oras[i].cam_states), len(oras[i].map_server)), (k, i)
            err = max(np.abs(out[i, 2:5] - s.position).max(), np.abs(out[i, 5:9] - s.orientation).max(), np.abs(out[i, 9:12] - s.velocity).max())
            assert err < 1e-6, (k, i, err)
            big = max(big, oras[i].state_cov.shape[0])
    assert big >= 21 + 6 * 23                                    # the window really grew to 23+ camera states
    for i in range(2):
        Po = oras[i].state_cov
        assert np.abs(bat.get_cov(i) - Po).max() <= 1e-6 * np.abs(Po).max()
    bat.close()


def test_one_streams_capacity_failure_does_not_stop_the_others(cfg):
    """stream 0 publishes 6,500 features per frame; its first camera-pruning update stacks more blocks
    than one update of the batched back end holds (4,096 blocks / the QR stage's row maps), which stops THAT stream (published = -1, stream_status says why).
    Stream 1 of the same group must keep following the oracle to the end."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd._native import AV_E_CAPACITY
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    n_frames = 40
    streams = [SyntheticFeatureStream(cfg, seed=91, n_frames=24, n_features=6500), SyntheticFeatureStream(cfg, seed=92, n_frames=n_frames, n_features=100)]
    bat = BatchedMSCKF(cfg, 2, max_features=6528)
    oras = [None, OracleMSCKF(cfg)]
    its = [iter(s.imu) for s in streams]; pend = [next(it, None) for it in its]
    failed_at = None
    for k in range(n_frames):
        msgs, out = _feed(bat, oras, streams, its, pend, k, 6528)
        if out[0, 0] < 0 and failed_at is None:
            failed_at = k
        if failed_at is not None and msgs[0] is not None:
            assert out[0, 0] == -1.0
        r = oras[1].feature_callback(msgs[1])
        assert (r is not None) == (out[1, 0] == 1.0), k
        if r is not None:
            s = oras[1].imu_state
            err = max(np.abs(out[1, 2:5] - s.position).max(), np.abs(out[1, 5:9] - s.orientation).max(), np.abs(out[1, 9:12] - s.velocity).max())
            assert err < 1e-6, (k, err)
            assert bat.sizes(1) == (oras[1].state_cov.shape[0], len(oras[1].cam_states), len(oras[1].map_server)), k
    assert failed_at is not None and 15 <= failed_at <= 23
    code, msg = bat.stream_status(0)
    assert code == AV_E_CAPACITY and ('blocks' in msg or 'QR stage' in msg), msg
    assert bat.stream_status(1) == (0, '')
    Po = oras[1].state_cov
    assert np.abs(bat.get_cov(1) - Po).max() <= 1e-6 * np.abs(Po).max()
    bat.close()


def test_degenerate_geometry_never_publishes_non_finite_poses(cfg):
    """long updates are compressed through the Cholesky factor of the Gram matrix Hc^T Hc, which squares
    the condition number of the stacked Jacobian.  Degenerate geometry is where that bites: a camera that stands still (zero
    parallax: every observation of a feature is a duplicate of the first, depth is unobservable from motion) and whole-window
    bursts of lost tracks (blank frames: hundreds of rows over ~100 columns at once).  Contract: a stream either publishes
    finite poses and a finite covariance, or it is stopped with a per-stream status (AV_E_NUMERIC / AV_E_CAPACITY) -- never
    NaN / Inf with status 0; and the well-conditioned stream of the same batch keeps following the oracle."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream, feature_msg_t
    n_frames = 70
    streams = [SyntheticFeatureStream(cfg, seed=71, n_frames=n_frames, n_features=150, motion_scale=0.0, pixel_sigma=0.0),      # static, noise-free: exact duplicates
               SyntheticFeatureStream(cfg, seed=72, n_frames=n_frames, n_features=150, motion_scale=1e-4, pixel_sigma=0.05),   # almost static
               SyntheticFeatureStream(cfg, seed=73, n_frames=n_frames, n_features=100)]                                          # ordinary
    S = len(streams)
    bat = BatchedMSCKF(cfg, S, rows_cap=8192)
    oras = [None, None, OracleMSCKF(cfg)]
    its = [iter(s.imu) for s in streams]; pend = [next(it, None) for it in its]
    blank = {0: (16, 30, 31, 52), 1: (18, 33, 55), 2: ()}
    cap = 192
    stopped = {}
    for k in range(n_frames):
        msgs = [s.frame(k) for s in streams]
        msgs = [feature_msg_t(m.timestamp, []) if k in blank[i] else m for i, m in enumerate(msgs)]
        si, ts, gy, ac = [], [], [], []
        for i, m in enumerate(msgs):
            while pend[i] is not None and pend[i].timestamp <= m.timestamp:
                if oras[i] is not None:
                    oras[i].imu_callback(pend[i])
                si.append(i); ts.append(pend[i].timestamp); gy.append(pend[i].angular_velocity); ac.append(pend[i].linear_acceleration)
                pend[i] = next(its[i], None)
        if si:
            bat.push_imu(si, ts, gy, ac)
        ids = np.zeros((S, cap), np.int64); uv = np.zeros((S, cap, 4)); nf = np.zeros(S, np.int32)
        for i, m in enumerate(msgs):
            nf[i] = len(m.features)
            for j, f in enumerate(m.features):
                ids[i, j] = f.id; uv[i, j] = (f.u0, f.v0, f.u1, f.v1)
        out = bat.step(ids, uv, nf, [m.timestamp for m in msgs])
        for i in range(S):
            if out[i, 0] < 0:
                stopped.setdefault(i, k)
                assert bat.stream_status(i)[0] != 0
                continue
            assert i not in stopped
            assert np.isfinite(out[i]).all(), (k, i, out[i])
            if out[i, 0] > 0.5:
                assert abs(np.linalg.norm(out[i, 5:9]) - 1.0) < 1e-9, (k, i)
        r = oras[2].feature_callback(msgs[2])
        assert (r is not None) == (out[2, 0] == 1.0), k
        if r is not None:
            s = oras[2].imu_state
            err = max(np.abs(out[2, 2:5] - s.position).max(), np.abs(out[2, 5:9] - s.orientation).max(), np.abs(out[2, 9:12] - s.velocity).max())
            assert err < 1e-6, (k, err)
    assert 2 not in stopped
    for i in range(S):
        if i not in stopped:
            assert bat.stream_status(i) == (0, '')
            P = bat.get_cov(i)
            assert np.isfinite(P).all() and (np.diag(P) > 0).all(), i
    print('\ndegenerate geometry: streams stopped by a status: %s' % ({i: (k, bat.stream_status(i)) for i, k in stopped.items()} or 'none'))
    bat.close()


def test_check_motion_with_a_positive_translation_threshold():
    """feature_motion_checker.py:6-39 is switched off in the reference's EuRoC configuration (translation_threshold = -1,
    config.py:12) but it is part of the feature_* surface: with a positive threshold a feature whose first and last observing
    cameras have not moved enough across its viewing direction is not triangulated -- dropped with the lost features
    (msckf.py:629-637), kept without a position by the camera pruning (:749-757).  The device-resident filter evaluates the test
    in `dk_append` / `dk_mid`; every frame must follow the numpy oracle, with features failing AND passing the test."""
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.config import ConfigEuRoC
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.synth import SyntheticFeatureStream
    cfg = ConfigEuRoC()
    cfg.optimization_config.translation_threshold = 0.04
    n_frames = 50
    streams = [SyntheticFeatureStream(cfg, seed=61, n_frames=n_frames, n_features=120), SyntheticFeatureStream(cfg, seed=62, n_frames=n_frames, n_features=60, motion_scale=0.5)]
    bat = BatchedMSCKF(cfg, 2)
    oras = [OracleMSCKF(cfg), OracleMSCKF(cfg)]
    # count how the oracle's check goes, to make sure both outcomes occur
    outcomes = {True: 0, False: 0}
    for o in oras:
        orig = o._check_motion

        def counted(feat, _orig=orig):
            r = bool(_orig(feat)); outcomes[r] += 1; return r
        o._check_motion = counted
    its = [iter(s.imu) for s in streams]; pend = [next(it, None) for it in its]
    for k in range(n_frames):
        msgs, out = _feed(bat, oras, streams, its, pend, k, 192)
        for i, m in enumerate(msgs):
            r = oras[i].feature_callback(m)
            assert (r is not None) == bool(out[i, 0]), (k, i)
            if r is None:
                continue
            s = oras[i].imu_state
            assert bat.sizes(i) == (oras[i].state_cov.shape[0], len(oras[i].cam_states), len(oras[i].map_server)), (k, i)
            err = max(np.abs(out[i, 2:5] - s.position).max(), np.abs(out[i, 5:9] - s.orientation).max(), np.abs(out[i, 9:12] - s.velocity).max())
            assert err < 1e-6, (k, i, err)
    assert outcomes[True] > 100 and outcomes[False] > 20, outcomes
    for i in range(2):
        Po = oras[i].state_cov
        assert np.abs(bat.get_cov(i) - Po).max() <= 1e-6 * np.abs(Po).max()
    bat.close()

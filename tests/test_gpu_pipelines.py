"""EngineSet / FilterSet (uav_airvision_amd/pipelines.py): P independent pipelines behind the interface of one give, per stream, exactly
what one engine + one filter give -- streams never interact."""
import numpy as np
import pytest


def test_split_parts_covers_the_streams_contiguously():
    from uav_airvision_amd.pipelines import split_parts
    assert split_parts(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert split_parts(4, 8) == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert split_parts(2048, 1) == [(0, 2048)]


@pytest.mark.gpu
def test_two_pipelines_publish_the_same_features_and_poses_as_one():
    import torch
    from uav_airvision_amd.config import ConfigEuRoC
    from uav_airvision_amd.frontend import FrontendEngine
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    from uav_airvision_amd.pipelines import EngineSet, FilterSet
    from uav_airvision_amd.synth import SyntheticStream
    from uav_airvision_amd import _native as N
    cfg = ConfigEuRoC()
    S, F = 4, 30
    streams = [SyntheticStream(cfg, seed=70 + s, n_frames=F) for s in range(S)]
    msgs = [[s.frame(k) for s in streams] for k in range(F)]
    img0 = [torch.from_numpy(np.stack([m.cam0_image for m in msgs[k]])).cuda().contiguous() for k in range(F)]
    img1 = [torch.from_numpy(np.stack([m.cam1_image for m in msgs[k]])).cuda().contiguous() for k in range(F)]
    ts = [[m.timestamp for m in msgs[k]] for k in range(F)]

    def imu_rows(k):
        idx, t, w, a = [], [], [], []
        for s, st in enumerate(streams):
            lo = ts[k - 1][s] if k else -1e18
            for m in st.imu:
                if lo < m.timestamp <= ts[k][s]:
                    idx.append(s); t.append(m.timestamp); w.append(m.angular_velocity); a.append(m.linear_acceleration)
        return np.array(idx, np.int32), np.array(t), np.array(w).reshape(-1, 3), np.array(a).reshape(-1, 3)

    def run(parts):
        if parts > 1:
            eng = EngineSet(cfg, S, parts, inputs_persist=True)
            flt = FilterSet(cfg, eng, max_features=eng.max_features)
        else:
            eng = FrontendEngine(cfg, n_streams=S, inputs_persist=True)
            flt = BatchedMSCKF(cfg, S, max_features=eng.max_features)
        assert flt.device_resident()
        feats, outs = [], []
        for k in range(F):
            i, t, w, a = imu_rows(k)
            eng.push_imu_batch(i, t, w)
            eng.step(img0[k], img1[k], ts[k])
            flt.push_imu(i, t, w, a)
            outs.append(flt.submit_dev(eng, np.asarray(ts[k]), msg_stream=N.current_stream()))
            feats.append(eng.read_features())
        flt.wait(0)
        poses = [np.array(o[:, :]) if not hasattr(o, 'array') else o.array() for o in outs]
        c = flt.counters()
        flt.close(); eng.close()
        return feats, poses, c

    f1, p1, c1 = run(1)
    f2, p2, c2 = run(2)
    for a, b in zip(f1, f2):
        for (ia, ua), (ib, ub) in zip(a, b):
            assert np.array_equal(ia, ib) and np.array_equal(ua.view(np.uint64), ub.view(np.uint64))
    assert sum(int(p[:, 0].sum() > 0) for p in p1) > 5                      # the filters did publish
    for a, b in zip(p1, p2):
        assert np.allclose(a, b, rtol=0, atol=1e-9), np.abs(a - b).max()
    assert c1['max_cam_states'] == c2['max_cam_states'] and c1['prune_stream_steps'] == c2['prune_stream_steps']

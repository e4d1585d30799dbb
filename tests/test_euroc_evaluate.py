"""Host-side 'next' rows: EuRoC reader + deterministic replay, trajectory format, ATE/RTE."""
import os

import numpy as np
import pytest

from uav_airvision_amd import euroc, evaluate


def _make_fake_euroc(root, n_img=6, imu_per_frame=10):
    from PIL import Image
    t0 = 1403636579763555584                      # ns, like the real sequences
    for cam in ('cam0', 'cam1'):
        d = os.path.join(root, 'mav0', cam, 'data'); os.makedirs(d)
        for k in range(n_img):
            img = np.full((480, 752), 10 * k + (0 if cam == 'cam0' else 1), np.uint8)
            Image.fromarray(img).save(os.path.join(d, '%d.png' % (t0 + k * 50_000_000)))
    os.makedirs(os.path.join(root, 'mav0', 'imu0'))
    with open(os.path.join(root, 'mav0', 'imu0', 'data.csv'), 'w') as f:
        f.write('#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z\n')
        for i in range(-20, n_img * imu_per_frame):
            f.write('%d,%f,%f,%f,%f,%f,%f\n' % (t0 + i * 5_000_000, 0.01 * i, 0.0, 0.1, 9.81, 0.0, 0.2))
    os.makedirs(os.path.join(root, 'mav0', 'state_groundtruth_estimate0'))
    with open(os.path.join(root, 'mav0', 'state_groundtruth_estimate0', 'data.csv'), 'w') as f:
        f.write('#timestamp,p,q,v,bw,ba\n')
        for i in range(0, n_img * imu_per_frame):
            f.write(','.join(['%d' % (t0 + i * 5_000_000)] + ['%f' % (0.001 * i * (c + 1)) for c in range(16)]) + '\n')
    return t0 * 1e-9


def test_reader_and_replay_order(tmp_path):
    t0 = _make_fake_euroc(str(tmp_path))
    ds = euroc.EuRoCDataset(str(tmp_path))
    assert len(ds.timestamps) == 6 and abs(ds.timestamps[0] - t0) < 1e-6
    # starttime = max(first imu, first image) = first image (IMU starts 0.1 s earlier)
    assert abs(ds.starttime0 - t0) < 1e-6
    events = []
    n = euroc.replay(ds, [lambda m: events.append(('imu', m.timestamp))], lambda m: events.append(('img', m.timestamp, int(m.cam0_image[0, 0]), int(m.cam1_image[0, 0]))))
    assert n == 6
    imgs = [e for e in events if e[0] == 'img']
    assert [e[2] for e in imgs] == [0, 10, 20, 30, 40, 50] and [e[3] for e in imgs] == [1, 11, 21, 31, 41, 51]
    # every IMU message delivered before a frame has timestamp <= that frame's time, and none is skipped
    last_img_t = -1.0
    for e in events:
        if e[0] == 'img':
            last_img_t = e[1]
        else:
            assert e[1] > last_img_t - 1e-9
    first_img = events.index(imgs[0])
    assert all(e[0] == 'imu' and e[1] <= imgs[0][1] + 1e-9 for e in events[:first_img]) and first_img == 1   # t0-0.1..t0 filtered by starttime
    m = next(iter(ds.stereo))
    assert m.cam0_msg.image.dtype == np.uint8 and m.cam0_msg.image.shape == (480, 752) and m.cam0_image is m.cam0_msg.image
    ds.set_starttime(0.12)                         # offset semantics of dataset.py:206-214
    assert [int(s.cam0_image[0, 0]) for s in ds.stereo] == [30, 40, 50]
    assert next(iter(ds.imu)).timestamp >= t0 + 0.12 - 1e-9
    g = ds.groundtruth_array()
    assert g.shape[1] == 8 and g[0, 0] >= t0 + 0.12 - 1e-9


def test_ate_recovers_known_alignment_and_noise():
    rng = np.random.default_rng(0)
    t = np.arange(0, 30, 0.05)
    gt = np.stack([t, np.sin(t), np.cos(0.5 * t), 0.1 * t], 1)
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec([0.2, -0.4, 1.0]).as_matrix()
    est = gt.copy()
    est[:, 1:4] = (gt[:, 1:4] - np.array([1.0, 2.0, 3.0])) @ R          # arbitrary frame
    a = evaluate.ate(est, gt)
    assert a['rmse'] < 1e-9 and a['n'] == len(t)
    est[:, 1:4] += rng.normal(0, 0.02, (len(t), 3))
    a = evaluate.ate(est, gt)
    assert 0.025 < a['rmse'] < 0.045                                     # sqrt(3) * 0.02
    r = evaluate.rte(est, gt, delta=10)
    assert 0.03 < r['rmse'] < 0.07                                       # sqrt(6) * 0.02
    est2 = est[::2].copy(); est2[:, 0] += 0.004                          # association tolerates small clock offsets
    assert evaluate.ate(est2, gt)['n'] == len(est2)


def test_trajectory_line_format_and_concatenated_runs(tmp_path):
    line = evaluate.format_state_line(1403636620.763556, [0.1, -0.2, 0.3], [0.0, 0.5, 0.0, 0.8660254])
    assert line == '1403636620.763556 0.100000000 -0.200000000 0.300000000 0.000000000 0.500000000 0.000000000 0.866025400\n'
    p = tmp_path / 'traj.txt'
    with open(p, 'w') as f:
        for tt in (5.0, 5.05, 5.1):                 # an earlier run left in the append-mode file (msckf.py:159)
            f.write(evaluate.format_state_line(tt, [9, 9, 9], [0, 0, 0, 1]))
        for tt in (1.0, 1.05, 1.1, 1.15):
            f.write(evaluate.format_state_line(tt, [tt, 0, 0], [0, 0, 0, 1]))
    a = evaluate.load_trajectory_txt(str(p))
    assert a.shape == (4, 8) and a[0, 0] == 1.0 and a[-1, 1] == 1.15


@pytest.mark.skipif(not os.path.isdir('/root/reference/results/txts'), reason='reference results only exist in the build container')
def test_parser_reads_the_reference_result_files():
    d = '/root/reference/results/txts'
    files = sorted(os.listdir(d))
    assert len(files) == 8
    a = evaluate.load_trajectory_txt(os.path.join(d, 'output_MH_01_easy_offset40.txt'))
    assert a.shape[1] == 8 and len(a) > 1000 and (np.diff(a[:, 0]) > 0).all()
    assert abs(np.linalg.norm(a[0, 4:8]) - 1) < 1e-6
    b = evaluate.load_trajectory_txt(os.path.join(d, 'output_MH_02_easy_offset30.txt'))   # two concatenated runs
    assert (np.diff(b[:, 0]) > 0).all() and len(b) < 3034

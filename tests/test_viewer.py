"""The no-op-compatible viewer object: the two hooks `modules/vio.py` calls plus `update_points`."""
import os
import sys
from collections import namedtuple

import numpy as np

from conftest import ROOT


def test_headless_viewer_records_what_the_reference_hooks_receive():
    d = os.path.join(ROOT, 'uav_airvision_amd', 'dropin')
    if d not in sys.path:
        sys.path.insert(0, d)
    from viewer import HeadlessViewer, SimpleViewer
    assert SimpleViewer is HeadlessViewer
    v = HeadlessViewer(history=3, keep_images=True)
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    v.update_image(img)
    img[0, 0] = 99                                       # the hook copies (viewer.py:49)
    assert v.n_images == 1 and v.last_image[0, 0] == 0
    pose = namedtuple('Iso', ['R', 't'])
    for k in range(5):
        v.update_pose(pose(np.eye(3), np.array([k, 2.0 * k, 0.5])))
    tr = v.trajectory()
    assert tr.shape == (3, 3) and np.array_equal(tr[:, 0], [2.0, 3.0, 4.0])      # bounded history, newest kept
    v.update_points(np.zeros((7, 3)))
    assert v.points.shape == (7, 3)
    v.close()
    v.update_pose(pose(np.eye(3), np.zeros(3)))           # after close the hooks are inert (viewer.py:46,52: `if self._running`)
    assert v.trajectory().shape == (3, 3)

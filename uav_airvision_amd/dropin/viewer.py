"""Headless counterpart of the reference's `viewer.SimpleViewer` hooks (reference: src/viewer.py:45-57).

`modules/vio.py` calls exactly two methods on the viewer it is given: `update_image(msg.cam0_image)` from the
image thread (`vio.py:32-33`) and `update_pose(result.cam0_pose)` from the filter thread (`vio.py:52-53`); the
reference's Qt window additionally offers `update_points`.  The reference class needs PyQt5, pyqtgraph, OpenGL and
a display; this one keeps the same three hooks with the same argument handling (the image is copied, the pose is
reduced to `T.t`) and only records, so `VIO(config, img_q, imu_q, viewer)` can be constructed and run on a
GPU box without a display.  Thread-safe the way the reference is: each hook only appends to its own container."""
import threading

import numpy as np


class HeadlessViewer(object):
    def __init__(self, history=1000, keep_images=False):
        self.history = int(history)
        self.keep_images = bool(keep_images)
        self._lock = threading.Lock()
        self._running = True
        self.last_image = None
        self.n_images = 0
        self.est_buf = np.empty((0, 3))                 # the same name as the reference's trajectory buffer (viewer.py:29)
        self.points = None

    def update_image(self, img):
        """viewer.py:45-49: keeps only the newest frame."""
        if self._running:
            with self._lock:
                self.n_images += 1
                self.last_image = np.array(img, copy=True) if self.keep_images else None

    def update_pose(self, T):
        """viewer.py:51-53: `T` is the `cam0_pose` Isometry3d of a `vio_result`; only its translation is used."""
        if self._running:
            t = np.asarray(T.t, dtype=np.float64).reshape(1, 3)
            with self._lock:
                self.est_buf = np.vstack([self.est_buf, t])[-self.history:]

    def update_points(self, pts):
        """viewer.py:55-57."""
        if self._running:
            with self._lock:
                self.points = np.asarray(pts)

    def trajectory(self):
        with self._lock:
            return self.est_buf.copy()

    def close(self):
        self._running = False


SimpleViewer = HeadlessViewer           # the reference's class name, for `from viewer import SimpleViewer`

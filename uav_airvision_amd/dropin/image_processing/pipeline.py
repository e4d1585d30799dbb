"""ImageProcessingPipeline on the MI355X engine (reference: src/image_processing/pipeline.py:14-150).

Same constructor argument and callbacks as the reference class; the work is done by one
device-resident `FrontendEngine` with a single stream (the throughput path batches S streams through
the same kernels, see uav_airvision_amd/frontend.py).  There is no CPU path.
"""
from collections import defaultdict, namedtuple

import numpy as np

from uav_airvision_amd.frontend import FrontendEngine

from .feature_measurment import FeatureMeasurement
from .feature_meta_data import FeatureMetaData

_feature_msg = namedtuple('feature_msg', ['timestamp', 'features'])


class ImageProcessingPipeline(object):
    def __init__(self, config, device=0, max_corners=8192):
        self.config = config
        self.prev_cam0_msg = None
        self._engine = FrontendEngine(config, n_streams=1, device=device, max_corners=max_corners)
        self.first_frame = True
        self.prev_pyr0 = None
        self.curr_features = [[] for _ in range(config.grid_num)]

    # ---- reference callbacks -----------------------------------------------------------------
    def imu_callback(self, imu_msg):
        """pipeline.py:42-44 -> imu_processor.py:22-26 (thread-safe against stereo_callback)."""
        self._engine.push_imu(0, imu_msg.timestamp, imu_msg.angular_velocity)

    def stereo_callback(self, stereo_msg):
        """pipeline.py:46-150: returns feature_msg(timestamp, [FeatureMeasurement])."""
        cam0_msg, cam1_msg = stereo_msg.cam0_msg, stereo_msg.cam1_msg
        img0 = np.ascontiguousarray(cam0_msg.image, dtype=np.uint8)
        img1 = np.ascontiguousarray(cam1_msg.image, dtype=np.uint8)
        if img0.ndim != 2 or img0.shape != (self._engine.height, self._engine.width) or img1.shape != img0.shape:
            raise ValueError('expected two uint8[%d,%d] images' % (self._engine.height, self._engine.width))
        self._engine.step_host(img0, img1, [cam0_msg.timestamp])
        (ids, uv), = self._engine.read_features()
        feats = []
        for k in range(len(ids)):
            fm = FeatureMeasurement()
            fm.id = int(ids[k])
            fm.u0, fm.v0, fm.u1, fm.v1 = uv[k, 0], uv[k, 1], uv[k, 2], uv[k, 3]
            feats.append(fm)
        self.prev_cam0_msg = cam0_msg
        self.prev_pyr0 = cam0_msg.image
        self.first_frame = False
        return _feature_msg(cam0_msg.timestamp, feats)

    # ---- pipeline state visible to callers (pipeline.py:33-40,145-148) -----------------------
    @property
    def prev_features(self):
        g = self._engine.read_grid(0)
        grid = [[] for _ in range(self.config.grid_num)]
        for k in range(len(g['ids'])):
            f = FeatureMetaData()
            f.id = int(g['ids'][k]); f.lifetime = int(g['lifetime'][k])
            f.cam0_point = g['cam0'][k]; f.cam1_point = g['cam1'][k]
            grid[int(g['cell'][k])].append(f)
        return grid

    @property
    def next_feature_id(self):
        return self._engine.read_grid(0)['next_feature_id']

    @property
    def num_features(self):
        c = self._engine.read_counters(0)
        d = defaultdict(int)
        d.update(before_tracking=c['before_tracking'], after_tracking=c['after_tracking'],
                 after_matching=c['after_matching'], after_ransac=c['after_matching'])
        return d

    def close(self):
        self._engine.close()

class FeatureMetaData(object):
    """Per-feature record of the front-end grid (reference: src/image_processing/feature_meta_data.py:1-10)."""

    def __init__(self):
        self.id = None
        self.response = None
        self.lifetime = None
        self.cam0_point = None
        self.cam1_point = None

class FeatureMeasurement(object):
    """Stereo measurement handed to the filter (reference: src/image_processing/feature_measurment.py:1-9)."""

    def __init__(self):
        self.id = None
        self.u0 = None
        self.v0 = None
        self.u1 = None
        self.v1 = None

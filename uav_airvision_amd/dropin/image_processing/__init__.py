"""`image_processing` package surface of the reference (src/image_processing/__init__.py:1-27)."""
from .pipeline import ImageProcessingPipeline
from .camera_model import CameraModel
from .imu_processor import IMUProcessor
from .pyramid_builder import PyramidBuilder
from .feature_meta_data import FeatureMetaData
from .feature_measurment import FeatureMeasurement
from .feature_stages import (FeatureInitializer, FeatureAdder, FeatureTracker, FeaturePruner, FeaturePublisher,
                             FastFeatureDetector_create)
from .stereo_matcher import StereoMatcher


class ImageProcessor(ImageProcessingPipeline):
    """Facade with the legacy alias (reference: src/image_processing/__init__.py:14-27)."""

    def __init__(self, config, **kw):
        super().__init__(config, **kw)

    stareo_callback = ImageProcessingPipeline.stereo_callback

"""`image_processing` package surface of the reference (src/image_processing/__init__.py:1-27)."""
from .pipeline import ImageProcessingPipeline
from .feature_meta_data import FeatureMetaData
from .feature_measurment import FeatureMeasurement


class ImageProcessor(ImageProcessingPipeline):
    """Facade with the legacy alias (reference: src/image_processing/__init__.py:14-27)."""

    def __init__(self, config, **kw):
        super().__init__(config, **kw)

    stareo_callback = ImageProcessingPipeline.stereo_callback

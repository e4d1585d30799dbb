"""Import path kept from the reference (the file name's spelling included): the type lives in records.py."""
from .records import FeatureMeasurement  # noqa: F401

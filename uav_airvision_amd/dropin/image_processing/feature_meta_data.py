"""Import path kept from the reference: the type lives in records.py."""
from .records import FeatureMetaData  # noqa: F401

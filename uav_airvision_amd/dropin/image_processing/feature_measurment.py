from .records import make_record

# stereo measurement handed to the filter: normalised cam0 / cam1 coordinates (reference: feature_measurment.py:1-9)
FeatureMeasurement = make_record('FeatureMeasurement', ('id', 'u0', 'v0', 'u1', 'v1'),
                                 'Stereo measurement handed to the filter (id, u0, v0, u1, v1).')

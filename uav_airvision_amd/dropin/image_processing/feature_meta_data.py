from .records import make_record

# per-feature record of the front-end grid (reference: feature_meta_data.py:1-10)
FeatureMetaData = make_record('FeatureMetaData', ('id', 'response', 'lifetime', 'cam0_point', 'cam1_point'),
                              'Per-feature record of the front-end grid (id, response, lifetime, cam0_point, cam1_point).')

"""The two record types the front-end stages exchange (reference attribute names kept so that callers reading
`.id`, `.u0` ... or `.cam0_point` ... keep working): plain attribute bags built from one field table each."""


def _record(name, fields, doc):
    def __init__(self, **kw):
        for f in fields:
            setattr(self, f, kw.pop(f, None))
        if kw:
            raise TypeError('%s: unknown field(s) %s' % (name, sorted(kw)))

    def __repr__(self):
        return '%s(%s)' % (name, ', '.join('%s=%r' % (f, getattr(self, f)) for f in fields))

    return type(name, (object,), {'__init__': __init__, '__repr__': __repr__, '__doc__': doc, '_fields': tuple(fields)})


# stereo measurement handed to the filter: normalised cam0 / cam1 coordinates (feature_measurment.py:1-9)
FeatureMeasurement = _record('FeatureMeasurement', ('id', 'u0', 'v0', 'u1', 'v1'),
                             'Stereo measurement handed to the filter (id, u0, v0, u1, v1).')
# per-feature record of the front-end grid (feature_meta_data.py:1-10)
FeatureMetaData = _record('FeatureMetaData', ('id', 'response', 'lifetime', 'cam0_point', 'cam1_point'),
                          'Per-feature record of the front-end grid (id, response, lifetime, cam0_point, cam1_point).')

"""Builder of the record types the front-end stages exchange: plain attribute bags made from one field table each (the reference's
attribute names are kept so that callers reading `.id`, `.u0` ... or `.cam0_point` ... keep working).  The two records live under
the reference's module names: feature_measurment.py (the spelling included) and feature_meta_data.py."""


def make_record(name, fields, doc):
    def __init__(self, **kw):
        for f in fields:
            setattr(self, f, kw.pop(f, None))
        if kw:
            raise TypeError('%s: unknown field(s) %s' % (name, sorted(kw)))

    def __repr__(self):
        return '%s(%s)' % (name, ', '.join('%s=%r' % (f, getattr(self, f)) for f in fields))

    return type(name, (object,), {'__init__': __init__, '__repr__': __repr__, '__doc__': doc, '_fields': tuple(fields)})

"""Parameter plumbing (reference: base_params.py:8-28; `bunch.Bunch` is an attribute dict)."""


class Bunch(dict):
    """Attribute-style dict standing in for the third-party `bunch.Bunch`."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return Bunch(self)


class BaseParams(object):
    """Every hot-path class exposes class_params() defaults and add_parse_options()."""

    @classmethod
    def class_params(cls):
        return Bunch()

    @classmethod
    def add_parse_options(cls, parser):
        pass

    @classmethod
    def get_updated_params(cls, options):
        """An option overrides a default only if the key exists AND the Python types match
        (base_params.py:24-27) -- this is why e.g. a dict-valued option never overrides."""
        params = cls.class_params()
        for attr in list(params.keys()):
            if attr in options and type(params[attr]) == type(options[attr]):
                params[attr] = options[attr]
        return params

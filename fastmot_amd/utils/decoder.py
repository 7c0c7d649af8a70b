"""JSON decoder that yields tuples for arrays -- config keys/values are API
(fastmot/utils/decoder.py:4-14, cfg/mot.json)."""
import json


def _tuplify(obj):
    if isinstance(obj, list):
        return tuple(_tuplify(v) for v in obj)
    if isinstance(obj, dict):
        return {k: _tuplify(v) for k, v in obj.items()}
    return obj


class ConfigDecoder(json.JSONDecoder):
    """`json.load(f, cls=ConfigDecoder, object_hook=...)` -> every JSON array becomes a tuple."""

    def __init__(self, **kwargs):
        self._user_hook = kwargs.pop('object_hook', None)
        super().__init__(object_hook=self._hook, **kwargs)

    def _hook(self, dct):
        dct = {k: _tuplify(v) for k, v in dct.items()}
        return self._user_hook(dct) if self._user_hook else dct

    def decode(self, s, **kwargs):
        return _tuplify(super().decode(s, **kwargs))

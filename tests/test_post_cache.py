"""The cache of post-process-only handles behind `postprocess.postprocess_global(params, arrays...)` (CPU: the handle class
is replaced by a recorder): keyed by the fields that shape a handle, bounded, least recently used evicted and closed."""
import numpy as np

from common import FULL_MC, make_params
from uda_amd import infer_lib, postprocess as PP


class _Fake:
    made, closed = [], []

    def __init__(self, name, n, only_network, params, post_only=False, chunk_images=1):
        self._cap, self.params = n, params
        _Fake.made.append(self)

    def close(self):
        _Fake.closed.append(self)


def test_cache_is_keyed_by_shaping_fields_and_bounded(monkeypatch):
    monkeypatch.setattr(infer_lib, "ServingDriver", _Fake)
    PP.close_cached()
    _Fake.made.clear(); _Fake.closed.clear()
    p = make_params(**FULL_MC)
    a = PP._post_driver(p, 2)
    assert PP._post_driver(dict(p, label_map={1: "car"}, mc_dropoutrate=0.2, moving_average_decay=0.5), 2) is a   # not shaping
    assert PP._post_driver(p, 1) is a                                   # a smaller batch fits the same handle
    b = PP._post_driver(p, 8)                                           # a larger one replaces it
    assert b is not a and a in _Fake.closed
    others = [PP._post_driver(dict(p, num_classes=c), 2) for c in (3, 4, 5)]
    assert len(PP._POST_DRIVERS) == 4 and b not in _Fake.closed
    PP._post_driver(p, 8)                                               # touch b: most recently used
    PP._post_driver(dict(p, num_classes=6), 2)                          # fifth configuration: evicts the LRU one = classes 3
    assert others[0] in _Fake.closed and b not in _Fake.closed and len(PP._POST_DRIVERS) == 4
    # stacking is shaping, the rate is not
    assert PP._cache_key(dict(p, mc_dropoutrate=0, mc_boxheadrate=0.1)) != PP._cache_key(p)
    assert PP._cache_key(dict(p, mc_dropoutsamp=7)) != PP._cache_key(p)
    n_alive = len(PP._POST_DRIVERS)
    PP.close_cached()
    assert not PP._POST_DRIVERS and len(_Fake.closed) == len(_Fake.made) and n_alive == 4

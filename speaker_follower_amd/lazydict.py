"""Result dictionaries whose expensive fields are built when they are read.

The reference's search and scoring procedures return lists of plain dictionaries (follower.py:694-716, 953-975;
speaker.py:139-141, 198-202) -- 2 500 of them per minibatch of 64 instructions in pragmatic inference, of which the
caller reads two or three scalar fields each (rational_follower.py:67-96) and deletes or ignores the rest.  A `LazyDict`
IS a dict (isinstance, json.dump, pickle, ==, iteration all work) holding the cheap fields; the names in `lazy` are
present too, and `_make(key)` builds one the first time it is read.  Anything that looks at the dictionary as a whole
builds what is missing first.
"""


class LazyDict(dict):
    __slots__ = ('_lazy',)

    def __init__(self, fields, lazy):
        dict.__init__(self, fields)
        self._lazy = list(lazy)

    def _make(self, key):
        raise NotImplementedError

    def __missing__(self, key):
        if key not in self._lazy:
            raise KeyError(key)
        self._lazy.remove(key)
        v = self._make(key)
        dict.__setitem__(self, key, v)
        return v

    def _all(self):
        for key in list(self._lazy):
            self[key]
        return self

    def __contains__(self, key):
        return key in self._lazy or dict.__contains__(self, key)

    def __setitem__(self, key, value):
        if key in self._lazy:
            self._lazy.remove(key)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        if key in self._lazy:
            self._lazy.remove(key)                     # (deleting a field nobody read builds nothing)
        else:
            dict.__delitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        return dict.pop(self._all(), key, *default)

    def setdefault(self, key, default=None):
        return self[key] if key in self else dict.setdefault(self, key, default)

    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v

    # whole-dictionary reads build what is missing first
    def __iter__(self):
        return dict.__iter__(self._all())

    def __len__(self):
        return dict.__len__(self._all())

    def __eq__(self, other):
        return dict.__eq__(self._all(), other._all() if isinstance(other, LazyDict) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        return dict.__repr__(self._all())

    def __reduce__(self):                              # (pickled / deep-copied as the plain dict it stands for)
        return dict, (dict(self._all()),)

    __hash__ = None

    def keys(self):
        return dict.keys(self._all())

    def values(self):
        return dict.values(self._all())

    def items(self):
        return dict.items(self._all())

    def copy(self):
        return dict(self._all())

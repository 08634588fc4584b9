"""Oracle (test infrastructure): numpy mirror of the counter-based dropout mask the
HIP kernels generate (speaker_follower_amd/csrc/sf_common.h: sf_dropout_keep).

The reference uses torch's stateful nn.Dropout (model.py:52, 370, 414, 473); its
random stream cannot be reproduced, so train-mode parity is checked by giving the
oracle the *same* mask the device derives from (seed, stream, row, col).
"""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = h & _M
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M
    h ^= h >> np.uint64(16)
    return h


def dropout_bits(seed, stream, rows, cols):
    """uint32 hash for every (row, col) pair: rows [R] (global row ids), cols [C]."""
    rows = np.asarray(rows, np.uint64)[:, None]
    cols = np.asarray(cols, np.uint64)[None, :]
    key = _fmix32(np.uint64(seed) + np.uint64(0x9E3779B9) * np.uint64(stream))
    key = _fmix32(key ^ ((rows * np.uint64(0x85EBCA6B)) & _M))
    return _fmix32(key + ((cols * np.uint64(0x9E3779B9)) & _M)).astype(np.uint32)


def dropout_mask(seed, stream, rows, n_cols, p):
    """Multiplicative mask [R, n_cols] fp32: 0 with probability p, else 1/(1-p)."""
    if p <= 0.0:
        return np.ones((len(rows), n_cols), np.float32)
    thresh = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    bits = dropout_bits(seed, stream, rows, np.arange(n_cols))
    keep = bits >= thresh
    return keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))

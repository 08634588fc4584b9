"""Parity bounds as north_star states them: logits within 1e-4 ABSOLUTE of the reference (fp32), actions / words
bit-exact.  Every check prints the measured maximum (pytest -s, and in the failure message), so a regression from
5e-5 to 5e-4 is visible long before it crosses the bound."""
import numpy as np

LOGIT_ATOL = 1e-4


def assert_logits_close(got, want, what, atol=LOGIT_ATOL, mask=None):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if mask is None:
        mask = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), mask), what + ': -inf pattern differs'
    d = float(np.abs(got[mask] - want[mask]).max())
    scale = float(np.abs(want[mask]).max())
    print('[parity] %s: max|dlogit| = %.3e at max|logit| = %.3f (bound %.0e absolute)' % (what, d, scale, atol))
    assert d <= atol, '%s: max|dlogit| = %.3e > %.0e (max|logit| = %.3f)' % (what, d, atol, scale)
    return d

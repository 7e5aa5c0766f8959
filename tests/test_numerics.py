"""is_logf (include/is_numerics.h) is the single logarithm of the device code and the oracle;
pin it to libm within 1 ulp and on the C99 special values."""
import numpy as np

from oracle import oracle


def _ulp_diff(a, b):
    ia = np.float32(a).view(np.int32).astype(np.int64)
    ib = np.float32(b).view(np.int32).astype(np.int64)
    return abs(int(ia) - int(ib))


def test_logf_specials():
    assert oracle.logf(1.0) == 0.0 and not np.signbit(np.float32(oracle.logf(1.0)))
    assert oracle.logf(0.0) == -np.inf and oracle.logf(-0.0) == -np.inf
    assert np.isnan(oracle.logf(-1.0)) and np.isnan(oracle.logf(float("nan")))
    assert oracle.logf(float("inf")) == np.inf


def test_logf_within_one_ulp_of_libm():
    rng = np.random.default_rng(11)
    xs = np.concatenate([
        rng.random(4000).astype(np.float32) * 2,
        (rng.random(3000) * 2000).astype(np.float32),
        np.exp(rng.uniform(-80, 80, 3000)).astype(np.float32),
        np.float32([0.3, 0.7, 2.0, 128.0, 1024.0, 1e-45, 1.1754944e-38, 3.4028235e38,
                    0.99999994, 1.0000001]),
        np.arange(1, 1100, dtype=np.float32),
    ])
    worst = 0
    for x in xs:
        want = np.log(np.float64(x)).astype(np.float32)
        worst = max(worst, _ulp_diff(oracle.logf(float(x)), want))
    assert worst <= 1

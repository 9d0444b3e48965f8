"""Arithmetic identities the HIP code relies on to replace a slow exact operation by a cheaper one with the
SAME bits.  Checked exhaustively on the CPU (fp32 fused multiply-add emulated exactly in float64)."""
import numpy as np


def fma32(a, b, c):
    # a*b has <= 48 significant bits and c 24: the sum is exact in float64 for the magnitudes used here
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def test_unorm8_by_multiply_and_residual_correction_equals_division_by_255():
    k = np.float32(1.0) / np.float32(255.0)
    for b in range(256):
        fb = np.float32(b)
        f = fb * k
        r = fma32(-f, np.float32(255.0), fb)
        assert fma32(r, k, f) == fb / np.float32(255.0), b
    # the plain multiply alone is NOT enough
    assert any(np.float32(b) * k != np.float32(b) / np.float32(255.0) for b in range(256))


def test_quantised_bound_product_is_exact():
    # q * step with q <= 255 and step a power of two is exact, so fma(q, step, origin) == origin + q * step
    rng = np.random.default_rng(0)
    for _ in range(2000):
        q = np.float32(rng.integers(0, 256))
        step = np.float32(2.0) ** np.float32(rng.integers(-30, 10))
        origin = np.float32(rng.normal() * 50)
        assert fma32(q, step, origin) == origin + q * step

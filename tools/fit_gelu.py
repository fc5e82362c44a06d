"""Coefficients of gelu_erf_fast (roma_amd/csrc/gemm_device.h): erfc(z / sqrt 2) = 2^-(z * Q(z)) with Q of degree 4.

    gelu(x) = x * Phi(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2)

-log2 erfc(z / sqrt 2) is smooth, zero at 0 and close to a parabola, so z * Q(z) with five coefficients reproduces it; the fit
minimises the ABSOLUTE error of the GELU value (weight = d gelu / d exponent) with a reweighted least squares that converges
towards the minimax solution.  `python tools/fit_gelu.py` prints the coefficients and the f32-evaluated error against erf;
tests/test_cpu_oracle.py checks the coefficients that are in the header with `eval_f32`.
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fit(degree=5, zmax=6.0, iters=200):
    from scipy.special import erfc
    z = np.linspace(1e-6, zmax, 60001)
    target = -np.log2(erfc(z / np.sqrt(2)))
    w = 0.5 * z * erfc(z / np.sqrt(2)) * np.log(2)
    A = np.vander(z, degree + 1, increasing=True)[:, 1:]
    best = None
    for _ in range(iters):
        co, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
        err = np.abs(0.5 * z * np.exp2(-(A @ co)) - 0.5 * z * erfc(z / np.sqrt(2)))
        if best is None or err.max() < best[0]:
            best = (err.max(), co.copy())
        w = w * (1 + err / err.max())
        w /= w.max()
    return best[1].astype(np.float32)


def eval_f32(x, co):
    """the header's instruction sequence in numpy float32 (fma rounding differences are far below the fit error)"""
    x = np.asarray(x, np.float32)
    z = np.abs(x)
    q = np.float32(co[-1])
    for c in co[-2::-1]:
        q = (q * z + np.float32(c)).astype(np.float32)
    e = np.exp2(-(q * z).astype(np.float32)).astype(np.float32)
    return (np.maximum(x, 0) + (np.float32(-0.5) * z) * e).astype(np.float32)


def coeffs_from_header():
    src = open(os.path.join(ROOT, "roma_amd", "csrc", "gemm_device.h")).read()
    body = src[src.index("float gelu_erf_fast(float x) {"):]
    body = body[:body.index("}")]
    c = [float(v) for v in re.findall(r"(-?\d\.\d+(?:e-?\d+)?)f[,)]", body) if v not in ("0.5", "-0.5", "0.")]
    # order in the source: c5, c4, c3, c2, c1
    return np.array(c[:5][::-1], np.float32)


if __name__ == "__main__":
    from scipy.special import erf
    co = fit()
    print("fit        :", [float(c) for c in co])
    print("in header  :", [float(c) for c in coeffs_from_header()])
    x = np.linspace(-9, 9, 2000001).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    for name, c in (("fit", co), ("header", coeffs_from_header())):
        err = np.abs(eval_f32(x, c) - ref)
        print(f"{name}: max |error| {err.max():.3e} at x = {x[err.argmax()]:.3f}")

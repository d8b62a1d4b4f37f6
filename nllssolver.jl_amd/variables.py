"""Host-side helpers for the registered variable kinds (construction only; the retraction
update() used during optimisation runs on the device, csrc/nlls_kinds.hpp)."""
import numpy as np


def so3_exp(w):
    """Rodrigues formula exp([w]x) (same series switch as the device code)."""
    w = np.asarray(w, dtype=np.float64)
    th2 = float(w @ w)
    if th2 < 1e-12:
        A, B = 1.0 - th2 / 6.0, 0.5 - th2 / 24.0
    else:
        th = np.sqrt(th2); A, B = np.sin(th) / th, (1.0 - np.cos(th)) / th2
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + A * Kx + B * (Kx @ Kx)


def contaminated_gaussian(s1, s2, w):
    """ContaminatedGaussian(s1, s2, w) -> storage (1/s1, 1/s2, w), narrowest Gaussian first
    (src/robustadaptive.jl:12-20)."""
    a, b = 1.0 / s1, 1.0 / s2
    if not a >= b:
        a, b = b, a
    return np.array([a, b, w])


def contaminated_gaussian_params(storage):
    """params(var)  src/robustadaptive.jl:23"""
    return np.array([1.0 / storage[0], 1.0 / storage[1], storage[2]])

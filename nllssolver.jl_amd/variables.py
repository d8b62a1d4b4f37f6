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


def contaminated_gaussian_em(storage, squarederrors, maxiters=10):
    """optimize(kernel::ContaminatedGaussian, squarederrors, maxiters)  src/robustadaptive.jl:48-73: the kernel's parameters by Expectation-Maximization
    on the blocks' squared errors (host side in the reference too: it is what the EM callback of test/adaptivecost.jl:15-25 calls between two iterations).
    storage = (1/sigma1, 1/sigma2, w); returns the new storage."""
    err = np.asarray(squarederrors, dtype=np.float64)
    k = np.array(storage, dtype=np.float64)
    total = float(err.sum())
    old = contaminated_gaussian_params(k)
    for _ in range(int(maxiters)):
        is1, is2, w = k
        wratio = ((1.0 - w) * is2) / (is1 * w)
        halfs1sqminuss2sq = -0.5 * (is2 * is2 - is1 * is1)
        with np.errstate(over="ignore"):                                          # (exp overflows to Inf for far outliers: their weight is then exactly 0, as in the reference)
            lat = 1.0 / (1.0 + wratio * np.exp(halfs1sqminuss2sq * err))       # expectation: the latent variables as a likelihood ratio
        sigma1 = float((lat * err).sum()); tw = float(lat.sum())
        new = np.array([np.sqrt(sigma1 / tw), np.sqrt((total - sigma1) / (err.size - tw)), tw / err.size])   # maximization
        k = contaminated_gaussian(*new)
        if np.linalg.norm(old - new) <= 1e-6 * max(np.linalg.norm(old), np.linalg.norm(new)):      # isapprox(oldparams, newparams; rtol=1.e-6)
            break
        old = new
    return k

"""Renyi-DP accountant for the sampled Gaussian mechanism (host math, numpy/scipy only).

Stands in for ``opacus.privacy_analysis.compute_rdp / get_privacy_spent`` which the reference calls
at mean_sampler.py:91-92 and, through ``privacy_engine.get_privacy_spent``, at train.py:295,588.
The Opacus fork is not in the container and is unpinned (requirements.txt:9): PARITY UNPINNED.
The formulas are the published ones — Mironov, Talwar, Zhang, "Renyi Differential Privacy of the
Sampled Gaussian Mechanism" (2019), Sec. 3.3 — and the RDP->(eps,delta) conversion of Balle et al.
(2020, Thm. 21) that Opacus 0.14 (the torch-1.9-era release the reference pins around) uses; the
classic conversion eps = rdp - log(delta)/(alpha-1) is available with ``improved=False``.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np
from scipy import special

DEFAULT_ALPHAS = [1 + x / 10.0 for x in range(1, 100)] + list(range(12, 400))  # train.py:99


def _log_add(a: float, b: float) -> float:
    lo, hi = min(a, b), max(a, b)
    if lo == -np.inf:
        return hi
    return math.log1p(math.exp(lo - hi)) + hi


def _log_sub(a: float, b: float) -> float:
    if a < b:
        raise ValueError("log-space subtraction would be negative")
    if b == -np.inf:
        return a
    if a == b:
        return -np.inf
    try:
        return math.log(math.expm1(a - b)) + b
    except OverflowError:
        return a


def _log_erfc(x: float) -> float:
    return math.log(2) + special.log_ndtr(-x * 2 ** 0.5)


def _log_a_int(q: float, sigma: float, alpha: int) -> float:
    log_a = -np.inf
    for i in range(alpha + 1):
        term = (math.log(special.binom(alpha, i)) + i * math.log(q) + (alpha - i) * math.log(1 - q)
                + (i * i - i) / (2 * sigma ** 2))
        log_a = _log_add(log_a, term)
    return float(log_a)


def _log_a_frac(q: float, sigma: float, alpha: float) -> float:
    log_a0, log_a1 = -np.inf, -np.inf
    z0 = sigma ** 2 * math.log(1 / q - 1) + 0.5
    i = 0
    while True:
        coef = special.binom(alpha, i)
        log_coef = math.log(abs(coef))
        j = alpha - i
        log_t0 = log_coef + i * math.log(q) + j * math.log(1 - q)
        log_t1 = log_coef + j * math.log(q) + i * math.log(1 - q)
        log_e0 = math.log(0.5) + _log_erfc((i - z0) / (math.sqrt(2) * sigma))
        log_e1 = math.log(0.5) + _log_erfc((z0 - j) / (math.sqrt(2) * sigma))
        log_s0 = log_t0 + (i * i - i) / (2 * sigma ** 2) + log_e0
        log_s1 = log_t1 + (j * j - j) / (2 * sigma ** 2) + log_e1
        if coef > 0:
            log_a0, log_a1 = _log_add(log_a0, log_s0), _log_add(log_a1, log_s1)
        else:
            log_a0, log_a1 = _log_sub(log_a0, log_s0), _log_sub(log_a1, log_s1)
        i += 1
        if max(log_s0, log_s1) < -30:
            break
    return _log_add(log_a0, log_a1)


def _rdp_one(q: float, sigma: float, alpha: float) -> float:
    if q == 0:
        return 0.0
    if sigma == 0:
        return np.inf
    if q == 1.0:
        return alpha / (2 * sigma ** 2)
    if np.isinf(alpha):
        return np.inf
    log_a = _log_a_int(q, sigma, int(alpha)) if float(alpha).is_integer() else _log_a_frac(q, sigma, alpha)
    return log_a / (alpha - 1)


def compute_rdp(q: float, noise_multiplier: float, steps: float, orders) -> np.ndarray:
    """RDP of `steps` compositions of the sampled Gaussian mechanism at each order."""
    if np.isscalar(orders):
        return _rdp_one(q, noise_multiplier, orders) * steps
    return np.array([_rdp_one(q, noise_multiplier, a) for a in orders]) * steps


def get_privacy_spent(orders, rdp, delta: float, improved: bool = True) -> Tuple[float, float]:
    """(epsilon, best order) for a target delta."""
    orders_vec, rdp_vec = np.atleast_1d(orders).astype(float), np.atleast_1d(rdp).astype(float)
    if len(orders_vec) != len(rdp_vec):
        raise ValueError("orders and rdp must have the same length")
    if improved:
        eps = rdp_vec - (np.log(delta) + np.log(orders_vec)) / (orders_vec - 1) + np.log((orders_vec - 1) / orders_vec)
    else:
        eps = rdp_vec - math.log(delta) / (orders_vec - 1)
    if np.isnan(eps).all():
        return np.inf, np.nan
    idx = int(np.nanargmin(eps))
    return float(eps[idx]), float(orders_vec[idx])

"""Effective sample size for many short chains (the reference has no ESS code; SURVEY.md §8d defines this one).

Per coordinate: FFT autocovariance per chain, combined across chains Stan-style (within-chain mean autocovariance +
between-chain variance), truncated with Geyer's initial positive / monotone sequence.  Host-side NumPy, used only for
reporting (bench.py, experiment harnesses)."""
import numpy as np


def _autocov(x):
    """x: (M, C) -> biased autocovariance (M, C) per chain via FFT."""
    M = x.shape[0]
    xc = x - x.mean(axis=0, keepdims=True)
    n = 1
    while n < 2 * M:
        n *= 2
    f = np.fft.rfft(xc, n=n, axis=0)
    ac = np.fft.irfft(f * np.conj(f), n=n, axis=0)[:M]
    return ac / M


def ess_detail(x):
    """x: (M, C) draws of one scalar for C chains -> dict with the total effective sample size `ess`, the integrated
    autocorrelation time `tau` (draws per effective draw, of the combined estimate), the lag `truncation_lag` at which
    Geyer's initial-positive sequence ended (== M means it never did: the estimate is then bounded by the window, not by
    the chain, and must not be quoted), and the split-free potential scale reduction `rhat` (sqrt(var_plus / W))."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    M, C = x.shape
    nan = float("nan")
    if M < 4:
        return {"ess": nan, "tau": nan, "truncation_lag": M, "rhat": nan, "draws": M, "chains": C}
    acov = _autocov(x)
    chain_var = acov[0] * M / (M - 1.0)
    W = chain_var.mean()
    var_plus = W * (M - 1.0) / M
    if C > 1:
        var_plus += x.mean(axis=0).var(ddof=1)
    if not np.isfinite(var_plus) or var_plus <= 0:
        return {"ess": nan, "tau": nan, "truncation_lag": M, "rhat": nan, "draws": M, "chains": C}
    rho = 1.0 - (W - acov.mean(axis=1)) / var_plus
    rho[0] = 1.0
    # Geyer: sums of adjacent pairs must be positive and non-increasing
    tau = -1.0
    prev = np.inf
    t = 0
    while t + 1 < M:
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        pair = min(pair, prev)
        prev = pair
        tau += 2.0 * pair
        t += 2
    lag = t if t + 1 < M else M
    tau = max(tau, 1.0 / np.log10(max(M * C, 10)))
    return {"ess": float(M * C / tau), "tau": float(tau), "truncation_lag": int(lag), "rhat": float(np.sqrt(var_plus / W)) if W > 0 else nan,
            "draws": M, "chains": C}


def ess(x):
    """x: (M, C) draws of one scalar for C chains -> total effective sample size (float)."""
    return ess_detail(x)["ess"]


def ess_ar1_exact(M, phi):
    """ESS of an AR(1) chain of length M (large M): M (1 - phi) / (1 + phi)."""
    return M * (1 - phi) / (1 + phi)

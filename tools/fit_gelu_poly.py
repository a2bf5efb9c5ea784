"""Fits the clamped odd polynomial of the 16-bit modes' FFN activation (amuse_dev.hpp gelu_poly4 / gelu_poly4h): erf(a / sqrt2) ~ a P(a^2),
a = clamp(x, +-X0), P of degree DEG, value at the clamp point pinned to 1.  DEG = 7, X0 = 3 sqrt2 is the bf16 kernel's (|erf error| 8.7e-5);
DEG = 10, X0 = 3.6 sqrt2 serves the fp16 mode (error below a quarter of an fp16 ulp).  Lawson-weighted least squares in float64, then the fp32
Horner evaluation is checked.  Build-container script: prints the coefficients pasted into the header.  Usage: fit_gelu_poly.py DEG X0/sqrt2"""
import sys
import numpy as np
from scipy.special import erf
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x0 = (float(sys.argv[2]) if len(sys.argv) > 2 else 3.6) * np.sqrt(2.0)
a = np.linspace(1e-6, x0, 20001)
s = a * a
y = erf(a / np.sqrt(2.0)) / a                       # P(s) target
# pin: P(x0^2) = 1 / x0  -> P(s) = 1/x0 + (s - x0^2) Q(s), Q of degree deg - 1
S0 = x0 * x0
t = (y - 1.0 / x0) / (s - S0 + 1e-300)
t[-1] = t[-2]
w = np.ones_like(a)
for _ in range(60):
    V = np.vander(s / S0, deg, increasing=True)      # scaled variable for conditioning
    W = np.sqrt(w)[:, None]
    c, *_ = np.linalg.lstsq(V * W * ((s - S0) * a)[:, None], (t * (s - S0) * a) * np.sqrt(w), rcond=None)
    e = np.abs((V @ c - t) * (s - S0) * a)
    w = w * (e / e.mean() + 1e-9); w /= w.sum()
q = c / S0 ** np.arange(deg)                         # Q in powers of s
p = np.zeros(deg + 1)
p[0] = 1.0 / x0
p[1:] += q
p[:-1] -= S0 * q                                     # P = 1/x0 + (s - S0) Q
p32 = p.astype(np.float32)
def ev(x):
    x = np.asarray(x, np.float32)
    aa = np.clip(x, -np.float32(x0), np.float32(x0)); ss = (aa * aa).astype(np.float32)
    r = np.full_like(ss, p32[-1])
    for ck in p32[-2::-1]:
        r = (r * ss + ck).astype(np.float32)
    return (aa * r).astype(np.float32)
xx = np.linspace(-8, 8, 400001).astype(np.float32)
err = np.abs(ev(xx).astype(np.float64) - erf(xx.astype(np.float64) / np.sqrt(2.0)))
print("X0 =", repr(float(np.float32(x0))), " max |erf error| fp32 eval:", err.max(), "at x =", xx[err.argmax()])
g = 0.5 * xx.astype(np.float64) * (1 + ev(xx).astype(np.float64)); gr = 0.5 * xx.astype(np.float64) * (1 + erf(xx.astype(np.float64) / np.sqrt(2)))
print("max |GELU error|:", np.abs(g - gr).max())
print("coefficients, highest power first:")
for ck in p32[::-1]:
    print("   %.9ef" % ck)

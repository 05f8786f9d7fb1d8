#!/usr/bin/env python3
"""Derives the coefficients of the fused-epilogue GELU in devit_amd/csrc/devit_common.h:
gelu(x) ~= x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)), x^2 clamped at 36, and reports the error of the value and of the
derivative against the erf form (models/de_vit.py:20, nn.GELU default) in float64."""
import numpy as np
from scipy.optimize import least_squares, minimize
from scipy.special import erf

x = np.linspace(-8, 8, 40001)
g = 0.5 * x * (1 + erf(x / np.sqrt(2)))


def approx(c, x):
    x2 = np.minimum(x * x, 36.0)
    return x / (1 + np.exp(-x * (c[0] + c[1] * x2 + c[2] * x2 * x2)))


c = least_squares(lambda c: approx(c, x) - g, [1.5957691, 0.0713548, 0.0]).x
for pw in (4, 8, 16):   # raise the norm towards minimax
    c = minimize(lambda c: np.sum(((approx(c, x) - g) * 1e4) ** pw), c, method="Nelder-Mead",
                 options=dict(xatol=1e-12, fatol=1e-14, maxiter=20000, maxfev=20000)).x
print("c0, c1, c2 =", c)
xx = np.linspace(-12, 12, 96001)
gg = 0.5 * xx * (1 + erf(xx / np.sqrt(2)))
print("max |gelu error|  =", np.abs(approx(c, xx) - gg).max())
x2 = np.minimum(xx * xx, 36.0)
s = 1 / (1 + np.exp(-xx * (c[0] + c[1] * x2 + c[2] * x2 * x2)))
dy = s + xx * (s - s * s) * (c[0] + 3 * c[1] * x2 + 5 * c[2] * x2 * x2)
dg = 0.5 * (1 + erf(xx / np.sqrt(2))) + xx * np.exp(-0.5 * xx * xx) / np.sqrt(2 * np.pi)
print("max |gelu' error| =", np.abs(dy - dg).max())

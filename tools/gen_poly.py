#!/usr/bin/env python3
"""Derive the fp32 polynomial coefficients used in manytor_amd/csrc/mt_math.h.

sin/cos are evaluated on a reduced argument f in [-45, 45] DEGREES (the quadrant
reduction f = x - 90*rint(x/90) is exact in fp32), atan on q in [0, 1].
Least-squares fit at Chebyshev nodes (close to minimax), printed as fp32 literals.
"""
import numpy as np

RAD = np.pi / 180


def nodes(n=4000):
    k = np.arange(n)
    return (np.cos(np.pi * (k + 0.5) / n) + 1) / 2


def fit(fun, deg, scale):
    w = nodes()
    a = np.vander(w, deg + 1, increasing=True)
    coef, *_ = np.linalg.lstsq(a, fun(w), rcond=None)
    return coef / (scale ** np.arange(deg + 1))


def sinq(w):
    f = 45 * np.sqrt(w)
    return np.where(f == 0, RAD, np.sin(f * RAD) / np.where(f == 0, 1, f))


def cosq(w):
    return np.cos(45 * np.sqrt(w) * RAD)


def atq(w):
    q = np.sqrt(w)
    return np.where(q == 0, 1.0, np.arctan(q) / np.where(q == 0, 1, q))


if __name__ == "__main__":
    for name, c in (("SIN_DEG", fit(sinq, 3, 2025.0)), ("COS_DEG", fit(cosq, 4, 2025.0)), ("ATAN", fit(atq, 7, 1.0))):
        print(name, ", ".join("%.9ef" % float(np.float32(v)) for v in c))

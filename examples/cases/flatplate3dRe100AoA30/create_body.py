#!/usr/bin/env python3
"""Lagrangian points of a flat plate of chord 1 and aspect ratio 2, inclined by 30 degrees, spacing 0.04 in the chord
and span directions (the discretisation of the reference's flatplate3dRe100_GPU case): writes flatplateAoA30.body in the
reference's body-file format (number of points, then one coordinate set per line)."""
import math
import os

import numpy as np

L, AR, aoa, ds = 1.0, 2.0, 30.0, 0.04
n = math.ceil(L / ds)
s = np.linspace(-L / 2, L / 2, num=n + 1)
x, y = np.cos(np.radians(-aoa)) * s, np.sin(np.radians(-aoa)) * s
nz = math.ceil(L * AR / ds)
z = np.linspace(-L * AR / 2, L * AR / 2, num=nz + 1)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flatplateAoA30.body")
with open(path, "w") as f:
    f.write(f"{x.size * z.size}\n")
    for zi in z:
        for xi, yi in zip(x, y):
            f.write(f"{xi:.18e} {yi:.18e} {zi:.18e}\n")
print(path, x.size * z.size, "points")

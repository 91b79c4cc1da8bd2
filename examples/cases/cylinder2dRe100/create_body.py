#!/usr/bin/env python3
"""158 Lagrangian points on a circle of radius 0.5 (the reference's circle.body), in the reference's body-file format."""
import os

import numpy as np

n = 158
a = 2.0 * np.pi * np.arange(n) / n
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "circle.body")
with open(path, "w") as f:
    f.write(f"{n}\n")
    for x, y in zip(0.5 * np.cos(a), 0.5 * np.sin(a)):
        f.write(f"{x:.18e} {y:.18e}\n")
print(path, n, "points")

#!/usr/bin/env python3
"""Two circles of radius 0.5 centred at (0, -2.5) and (0, 2.5), 158 Lagrangian points each (arc length 0.02), in the
reference's body-file format (circle1.body, circle2.body of its multicylinders2dRe100_GPU example)."""
import os

import numpy as np

n = int(np.ceil(2.0 * np.pi * 0.5 / 0.02))
a = 2.0 * np.pi * np.arange(n) / n
here = os.path.dirname(os.path.abspath(__file__))
for name, yc in (("circle1.body", -2.5), ("circle2.body", 2.5)):
    with open(os.path.join(here, name), "w") as f:
        f.write(f"{n}\n")
        for x, y in zip(0.5 * np.cos(a), yc + 0.5 * np.sin(a)):
            f.write(f"{x:.18e} {y:.18e}\n")
    print(name, n, "points")

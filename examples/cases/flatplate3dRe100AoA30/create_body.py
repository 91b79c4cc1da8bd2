#!/usr/bin/env python3
"""Markers of an inclined flat plate, from the plate's definition: a rectangle of chord 1 (along the flow before it is
pitched) and span 2 (along z), centred at the origin, covered by a lattice of markers one marker spacing (0.04, the mesh
width around the body in config.yaml) apart in both directions, then pitched nose-up by 30 degrees about the span axis.
Writes flatplateAoA30.body in the body-file format the flow engine reads (petibm_amd.navierstokes.read_lagrangian_points:
the number of markers, then one "x y z" line per marker)."""
import os

import numpy as np

CHORD, SPAN, PITCH_DEG, SPACING = 1.0, 2.0, 30.0, 0.04


def lattice(length: float, spacing: float) -> np.ndarray:
    """marker stations across `length`, centred, both edges included, no further apart than `spacing`"""
    intervals = int(np.ceil(length / spacing - 1e-12))
    return (np.arange(intervals + 1) / intervals - 0.5) * length


def plate_markers() -> np.ndarray:
    along_chord, along_span = lattice(CHORD, SPACING), lattice(SPAN, SPACING)
    # the flat plate in its own frame: (chord station, 0, span station), chord stations running fastest
    c, z = np.meshgrid(along_chord, along_span, indexing="xy")
    flat = np.stack([c.ravel(), np.zeros(c.size), z.ravel()], axis=1)
    # nose-up pitch = a rotation by -PITCH about z (the leading edge, at negative chord stations, rises)
    t = np.radians(-PITCH_DEG)
    rot = np.array([[np.cos(t), -np.sin(t), 0.0], [np.sin(t), np.cos(t), 0.0], [0.0, 0.0, 1.0]])
    return flat @ rot.T


if __name__ == "__main__":
    pts = plate_markers()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flatplateAoA30.body")
    np.savetxt(path, pts, fmt="%.18e", header=str(len(pts)), comments="")
    print(path, len(pts), "points")

#!/usr/bin/env python3
"""Provenance of tests/golden/reference_test_vectors.json.

The fixture holds DATA the reference ships for this path -- nothing of its source code:

* transcribed by hand from the expected values in the reference's unit tests (not regenerable by a script; the
  `source` field of each entry names file and lines): cartesianmesh2d_dirichlet, cartesianmesh2d_yperiodic,
  cartesianmesh3d_dirichlet, createbnhead, delta_roma_et_al_1999, and the README / documentation constants
  (multicylinders2dRe100_readme, taira...petibm_figure_at_30deg, taylor_green_vortex_2d_re100_readme);
* extracted from the data files under /root/reference/examples (this script): the Ghia et al. centre-line tables,
  the Koumoutsakos & Leonard drag histories, the Taira et al. force coefficients, the spectral Taylor-Green energies.

Run in the build container (the only place /root/reference exists) to check that the committed fixture still equals
what the data files say:    python tests/golden/extract_reference_data.py [--write]
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference/examples"
HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "reference_test_vectors.json")


def extracted():
    out = {}
    g = np.loadtxt(f"{REF}/data/ghia_et_al_1982_lid_driven_cavity.dat")
    out["ghia_1982_re100_u_centerline"] = {"y": g[:, 0], "u": g[:, 1], "x": g[:, 6], "v": g[:, 7]}
    for re, col in ((1000, 2), (3200, 3), (5000, 4)):
        out[f"ghia_1982_re{re}_centerlines"] = {"y": g[:, 0], "u": g[:, col], "x": g[:, 6], "v": g[:, 6 + col]}
    for re in (40, 550, 3000):
        a = np.loadtxt(f"{REF}/data/koumoutsakos_leonard_1995_cylinder_dragCoefficientRe{re}.dat")
        out[f"koumoutsakos_leonard_1995_cylinder_re{re}"] = {"t_radius_units": a[:, 0], "cd": a[:, 1]}
    cd = np.loadtxt(f"{REF}/data/taira_et_al_2007_flatPlateRe100AR2_CdvsAoA.dat")
    cl = np.loadtxt(f"{REF}/data/taira_et_al_2007_flatPlateRe100AR2_ClvsAoA.dat")
    out["taira_et_al_2007_flatplate_re100_ar2"] = {"cd_aoa": cd[:, 0], "cd": cd[:, 1], "cl_aoa": cl[:, 0], "cl": cl[:, 1]}
    s = np.loadtxt(f"{REF}/navierstokes/taylorgreenvortex3dRe1600_GPU/resources/spectral_Re1600_512.gdiag")
    rows = []
    for tt in np.arange(0.0, 19.9, 0.5):
        i = int(np.argmin(np.abs(s[:, 0] - tt)))
        rows.append([round(float(s[i, 0]), 6), float(s[i, 1]), float(s[i, 2])])
    out["taylor_green_vortex_3d_re1600_spectral_512"] = {"rows": rows}
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("the reference is not mounted here; nothing to check")
    G = json.load(open(PATH))
    bad = 0
    for key, fields in extracted().items():
        for name, val in fields.items():
            have = G.get(key, {}).get(name)
            want = np.asarray(val, dtype=float)
            if have is None or np.asarray(have, dtype=float).shape != want.shape or not np.array_equal(np.asarray(have, dtype=float), want):
                bad += 1
                print(f"differs: {key}.{name}")
                if "--write" in sys.argv:
                    G.setdefault(key, {})[name] = want.tolist()
    if "--write" in sys.argv and bad:
        json.dump(G, open(PATH, "w"), indent=1)
        print("fixture updated")
    print("fixture equals the reference's data files" if not bad else f"{bad} field(s) differ")
    return 0 if (not bad or "--write" in sys.argv) else 1


if __name__ == "__main__":
    sys.exit(main())

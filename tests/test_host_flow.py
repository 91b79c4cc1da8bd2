"""Host-side logic of the flow-solver mirrors (no GPU): mesh arithmetic, initial conditions, body files -- Python
(petibm_amd/navierstokes.py) and C++ (include/petibm_amd/flowsolver.hpp) against the oracle's restatement of the
reference's parser (src/parser/parser.cpp:298-356, include/petibm/misc.h:148-163, src/io/io.cpp:23-118)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import mesh as omesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AXIS = {"direction": "x", "start": -15.0, "subDomains": [{"end": -0.54, "cells": 171, "stretchRatio": 0.980392156},
                                                          {"end": 0.54, "cells": 108, "stretchRatio": 1.0},
                                                          {"end": 15.0, "cells": 171, "stretchRatio": 1.02}]}


def test_python_mirror_mesh_arithmetic_is_the_oracle_s():
    from petibm_amd import navierstokes as pn
    w, lo, hi = pn._widths(AXIS)
    n, end, want = omesh.parse_subdomains(AXIS["subDomains"], AXIS["start"])
    assert n == len(w) == 450 and lo == -15.0 and hi == end == 15.0
    assert np.array_equal(w, want)  # same operations in the same order: identical bits


def test_python_mirror_initial_velocity_expressions():
    from petibm_amd import navierstokes as pn

    class Bare(pn.NavierStokesSolver):
        def __init__(self):  # no device
            self.dim, self.nu = 2, 0.1
            self.widths = [np.full(4, 0.25), np.full(3, 1.0 / 3.0)]
            self.periodic = [False, False]

    s = Bare()
    U = s._initial_velocity(["sin(x)*cos(y)", 2.0], [0.0, 0.0])
    xu, yu = np.array([0.25, 0.5, 0.75]), (np.arange(3) + 0.5) / 3.0
    want_u = (np.sin(xu)[None, :] * np.cos(yu)[:, None]).reshape(-1)
    assert U.shape == (3 * 3 + 4 * 2,) and np.allclose(U[:9], want_u, rtol=1e-15) and np.all(U[9:] == 2.0)
    V = s._initial_velocity(["exp(-2*nu*t)*x^2", "pi"], [0.0, 0.0])   # '^' as in the SymEngine syntax of the reference
    assert np.allclose(V[:3], xu ** 2) and np.allclose(V[9:], np.pi)
    with pytest.raises(Exception):
        s._initial_velocity(["__import__('os').system('true')", 0.0], [0.0, 0.0])  # no builtins in the evaluator
    # periodic x: one more u point per line, at the + boundary (cartesianmesh.cpp:249-266); pressure expression
    s.periodic = [True, False]
    W = s._initial_velocity(["x", 0.0], [0.0, 0.0])
    assert W.shape == (4 * 3 + 4 * 2,) and np.allclose(W[:4], [0.25, 0.5, 0.75, 1.0])
    P = s._initial_velocity(["x + y"], [0.0, 0.0], pressure=True)
    assert P.shape == (12,) and np.isclose(P[0], 0.125 + 1.0 / 6.0)


def test_body_files(tmp_path):
    from petibm_amd import capi, navierstokes as pn
    pts = np.random.default_rng(0).uniform(-1, 1, (5, 3))
    f = tmp_path / "b.body"
    f.write_text("5\n" + "\n".join(" ".join(f"{v:.18e}" for v in p) for p in pts) + "\n")
    assert np.array_equal(pn.read_lagrangian_points(str(f)), pts)
    f.write_text("4\n" + "\n".join(" ".join(f"{v:.18e}" for v in p) for p in pts) + "\n")
    with pytest.raises(capi.PibError) as ei:
        pn.read_lagrangian_points(str(f))
    assert ei.value.code == capi.ERR_FILE_READ
    f.write_text("5 3\n")
    with pytest.raises(capi.PibError):
        pn.read_lagrangian_points(str(f))


def test_cpp_mirror_mesh_arithmetic_and_body_reader(tmp_path, built_library):
    """flowsolver.hpp compiled with plain g++: cellWidths (parseSubDomains + stretchGrid) and readLagrangianPoints give
    the oracle's numbers (printed with 17 significant digits, compared exactly)."""
    src = tmp_path / "t.cpp"
    body = tmp_path / "c.body"
    body.write_text("3\n0.5 0.25\n-0.125 1e-3\n7 8\n")
    src.write_text(r'''
#include <cstdio>
#include "petibm_amd/flowsolver.hpp"
using namespace petibm_amd;
int main(int argc, char **argv) {
    MeshAxis a{-15.0, {{-0.54, 171, 0.980392156}, {0.54, 108, 1.0}, {15.0, 171, 1.02}}};
    double end = 0;
    std::vector<double> w = cellWidths(a, &end);
    std::printf("%zu %.17g\n", w.size(), end);
    for (double v : w) std::printf("%.17g\n", v);
    std::vector<double> c; int64_t n = 0;
    int e = readLagrangianPoints(argv[1], 2, c, n);
    std::printf("body %d %lld %.17g %.17g\n", e, (long long)n, c[2], c[5]);
    std::printf("missing %d\n", readLagrangianPoints("/nonexistent.body", 2, c, n));
    return 0;
}
''')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L",
                           os.path.dirname(built_library), "-lpetibm_amd", f"-Wl,-rpath,{os.path.dirname(built_library)}",
                           "-o", str(exe)])
    out = subprocess.run([str(exe), str(body)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split("\n")
    n, end = lines[0].split()
    want = omesh.parse_subdomains(AXIS["subDomains"], AXIS["start"])[2]
    got = np.array([float(v) for v in lines[1:1 + int(n)]])
    assert int(n) == 450 and float(end) == 15.0 and np.array_equal(got, want)
    assert lines[1 + int(n)] == "body 0 3 -0.125 8" and lines[2 + int(n)] == "missing 65"


def test_xdmf_descriptions(tmp_path):
    """petibm_amd/xdmf.py: one well-formed .xmf per field, a temporal collection over the saved steps, the point counts
    of the staggered fields (createxdmf/main.cpp)"""
    import xml.dom.minidom
    from petibm_amd import xdmf
    paths = xdmf.write_all(str(tmp_path), 2, (32, 24), (False, True), [0, 100, 200])
    assert sorted(os.path.basename(p) for p in paths) == ["p.xmf", "u.xmf", "v.xmf", "wz.xmf"]
    for p in paths:
        text = open(p).read()
        xml.dom.minidom.parseString(text)  # entities are internal: a standard parser accepts the file
        assert text.count("<Time Value=") == 3 and "0000200.h5" in text
    u, v = open(paths[0]).read(), open(paths[1]).read()
    assert '<!ENTITY Nx "31">' in u and '<!ENTITY Ny "24">' in u
    assert '<!ENTITY Nx "32">' in v and '<!ENTITY Ny "24">' in v   # periodic y: one more line of v points
    assert "Format='XML'" in u  # the dummy z axis of a 2-D run

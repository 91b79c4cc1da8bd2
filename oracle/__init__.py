"""CPU oracle for the PetIBM linear-solve hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT.  It is a CPU restatement
(numpy + plain C) of the algorithms the reference runs on the path

    mesh arithmetic -> operators (D, G, L, A, BN, DBNG) -> Krylov solve

and exists only so that the hand-written HIP path in ``petibm_amd`` can be
checked against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``petibm_amd/`` imports, links or executes anything from here.

Pinning status (see DESIGN.md "Oracle"):
  * mesh arithmetic  -- PINNED against the reference's own golden vectors
    (tests/mesh/cartesianmesh2d_dirichlet.cpp:171-285,
     tests/mesh/cartesianmesh2d_yperiodic.cpp:160-285,
     tests/mesh/cartesianmesh3d_dirichlet.cpp:81-108), committed as data under
    tests/golden/.
  * createBnHead     -- PINNED against the known-answer test
    tests/operators/createbnhead_test.cpp:17-61.
  * Laplacian / divergence / gradient entries and every LinSolver::solve --
    "parity unpinned": the reference has no test, golden vector or fixture for
    them, and the reference cannot be built in this image (it needs PETSc 3.16,
    yaml-cpp, SymEngine -- none present, no network).  The Krylov arithmetic
    itself lives in PETSc 3.16 (KSPCG/KSPBCGS/PCJACOBI) and AmgX 2.2.0 (PCG,
    PBICGSTAB), both third-party and absent; their published recurrences are
    restated in oracle/csrc/oracle.c and cross-checked against scipy.sparse in
    tests/ (scipy is a second, independent implementation, not the reference).
"""

"""XDMF descriptions of the HDF5 output for VisIt / ParaView, in the layout the reference's `petibm-createxdmf` utility
writes (applications/createxdmf/main.cpp:60-127 which fields, :129-268 one file per field): a temporal collection of
uniform grids, every grid a 3-D rectilinear mesh (a 2-D run gets a dummy z axis) whose gridlines live in grid.h5 and
whose values live in <step>.h5.  The entity names (CaseDir, Nx, Ny, Nz, Topo, Geo) are the reference's, so files
written here and there are interchangeable."""
from __future__ import annotations

import os


def write_single_xdmf(directory: str, name: str, dim: int, n, steps) -> str:
    """<directory>/<name>.xmf for the dataset `name` with n = (nx, ny, nz) points saved at the time steps `steps`."""
    n = list(n) + [1] * (3 - len(n))
    axes = "xyz"
    geo = ["\t<!ENTITY Geo", "\t\t\"<Geometry GeometryType='VXVYVZ'>"]
    for d in range(dim):
        geo += [f"\t\t\t<DataItem Dimensions='&N{axes[d]};' Format='HDF' Precision='8'>",
                f"\t\t\t\t&CaseDir;/grid.h5:/{name}/{axes[d]}", "\t\t\t</DataItem>"]
    if dim == 2:
        geo += ["\t\t\t<DataItem Dimensions='&Nz;' Format='XML' Precision='8'>", "\t\t\t\t0.0", "\t\t\t</DataItem>"]
    geo += ["\t\t</Geometry>\"", "\t>"]
    out = ["<?xml version='1.0' ?>", "", "<!DOCTYPE Xdmf SYSTEM \"Xdmf.dtd\" [", "\t<!ENTITY CaseDir \"./\">"]
    out += [f"\t<!ENTITY N{axes[d]} \"{int(n[d])}\">" for d in range(3)]
    out += ["\t<!ENTITY Topo \"<Topology TopologyType='3DRectMesh' Dimensions='&Nz; &Ny; &Nx;'/>\">"]
    out += geo + ["]>", "", "<Xdmf Version=\"3.0\">", "\t<Domain>", "\t<Grid GridType=\"Collection\" CollectionType=\"Temporal\">"]
    for t in steps:
        out += [f"\t\t<Grid GridType=\"Uniform\" Name=\"{name} Grid\">", f"\t\t\t<Time Value=\"{int(t):07d}\" />",
                "\t\t\t&Topo; &Geo;", f"\t\t\t<Attribute Name=\"{name}\" AttributeType=\"Scalar\" Center=\"Node\">",
                "\t\t\t\t<DataItem Dimensions=\"&Nz; &Ny; &Nx;\" Format=\"HDF\" NumberType=\"Float\" Precision=\"8\">",
                f"\t\t\t\t\t&CaseDir;/{int(t):07d}.h5:/{name}", "\t\t\t\t</DataItem>", "\t\t\t</Attribute>", "\t\t</Grid>"]
    out += ["\t</Grid>", "\t</Domain>", "</Xdmf>", ""]
    path = os.path.join(directory, name + ".xmf")
    with open(path, "w") as f:
        f.write("\n".join(out))
    return path


def write_all(directory: str, dim: int, cells, periodic, steps, vorticity: bool = True):
    """u, v[, w], p and (optionally) the vorticity fields, with the point counts of each (createxdmf/main.cpp:60-127;
    mesh->n of cartesianmesh.cpp:136-355)"""
    cells = list(cells)
    paths = []
    for f, name in enumerate("uvw"[:dim]):
        n = [cells[d] - (1 if (d == f and not periodic[d]) else 0) for d in range(dim)]
        paths.append(write_single_xdmf(directory, name, dim, n, steps))
    paths.append(write_single_xdmf(directory, "p", dim, cells, steps))
    if vorticity:
        for comp in ([2] if dim == 2 else [0, 1, 2]):
            n = [cells[d] + (0 if d == comp else 1) for d in range(dim)]
            paths.append(write_single_xdmf(directory, "w" + "xyz"[comp], dim, n, steps))
    return paths

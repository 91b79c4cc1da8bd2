"""The ctypes HDF5 layer (petibm_amd/h5io.py) that writes / reads the reference's solution, grid and restart files."""
import numpy as np
import pytest

h5io = pytest.importorskip("petibm_amd.h5io")
try:
    h5io.lib()
except ImportError as e:  # the image ships libhdf5 under /opt/conda; elsewhere the feature is simply absent
    pytest.skip(str(e), allow_module_level=True)


def test_datasets_groups_attributes_round_trip(tmp_path):
    p = str(tmp_path / "0000100.h5")
    u = np.random.default_rng(0).uniform(-1, 1, (5, 4))
    w3 = np.arange(24.0).reshape(2, 3, 4)
    with h5io.File(p, "w") as f:
        f.write("u", u)
        f.write("p", np.ones((5, 5)))
        f.write_attr("p", "time", 0.125)
        f.write("w", w3)
    with h5io.File(p, "a") as f:  # FILE_MODE_APPEND of writeRestartDataHDF5
        f.write("convection/0", np.arange(7.0))
        f.write("convection/1", -np.arange(7.0))
        f.write("force", np.array([1.5]))
        f.write("u", 2.0 * u)       # same name: replaced
        f.write_attr("p", "time", 0.25)
    with h5io.File(p, "r") as f:
        assert np.array_equal(f.read("u"), 2.0 * u) and f.read("w").shape == (2, 3, 4) and np.array_equal(f.read("w"), w3)
        assert f.read_attr("p", "time") == 0.25
        assert np.array_equal(f.read("convection/1"), -np.arange(7.0)) and f.exists("force") and not f.exists("diffusion/0")
        with pytest.raises(h5io.H5Error):
            f.read("nothing")
    with pytest.raises(h5io.H5Error):
        h5io.File(str(tmp_path / "missing.h5"), "r")

"""The RCCL call forms csrc/halo.hip uses (the only lines the loopback transport does not cover), run on a
ONE-rank communicator of the RCCL this process loads: id + ncclCommInitRank, in-place ncclAllReduce of doubles,
in-place ncclAllGather of int64, a ncclGroup of ncclSend/ncclRecv (to self) on a non-default stream, a group of
ncclBroadcast.  It proves the library, its symbols and the stream usage on this box -- not the multi-rank data path
(RCCL refuses several ranks per device; the loopback tests cover that logic)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_call_forms_single_rank():
    import torch
    from petibm_amd import capi
    lib = capi.load()
    uid = C.create_string_buffer(capi.UID_BYTES)
    capi.check(lib.pib_comm_unique_id(uid))  # ncclGetUniqueId through the product entry point
    assert any(uid.raw), "empty RCCL id"

    rccl = C.CDLL("librccl.so.1")  # SONAME of the copy already in the process (torch's)
    vp, sz = C.c_void_p, C.c_size_t

    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]

    rccl.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, Uid, C.c_int]
    rccl.ncclAllReduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
    rccl.ncclAllGather.argtypes = [vp, vp, sz, C.c_int, vp, vp]
    rccl.ncclSend.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp]
    rccl.ncclRecv.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp]
    rccl.ncclBroadcast.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
    rccl.ncclCommDestroy.argtypes = [vp]
    ncclInt64, ncclDouble, ncclSum = 4, 8, 0

    torch.cuda.set_device(0)
    u = Uid()
    C.memmove(C.byref(u), uid, 128)
    comm = vp()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0
    st = torch.cuda.Stream()
    s = vp(st.cuda_stream)
    with torch.cuda.stream(st):
        d = torch.arange(8, dtype=torch.float64, device="cuda")
        assert rccl.ncclAllReduce(d.data_ptr(), d.data_ptr(), 8, ncclDouble, ncclSum, comm, s) == 0
        g = torch.tensor([5, 6, 7, 8], dtype=torch.int64, device="cuda")
        assert rccl.ncclAllGather(g.data_ptr(), g.data_ptr(), 4, ncclInt64, comm, s) == 0
        x = torch.arange(16, dtype=torch.float64, device="cuda")
        assert rccl.ncclGroupStart() == 0
        assert rccl.ncclSend(x.data_ptr(), 4, ncclDouble, 0, comm, s) == 0
        assert rccl.ncclRecv(x.data_ptr() + 8 * 8, 4, ncclDouble, 0, comm, s) == 0
        assert rccl.ncclGroupEnd() == 0
        y = torch.arange(6, dtype=torch.float64, device="cuda")
        assert rccl.ncclGroupStart() == 0
        assert rccl.ncclBroadcast(y.data_ptr(), y.data_ptr(), 6, ncclDouble, 0, comm, s) == 0
        assert rccl.ncclGroupEnd() == 0
    st.synchronize()
    assert np.array_equal(d.cpu().numpy(), np.arange(8.0))
    assert g.cpu().tolist() == [5, 6, 7, 8]
    xe = np.arange(16.0)
    xe[8:12] = xe[0:4]
    assert np.array_equal(x.cpu().numpy(), xe)
    assert rccl.ncclCommDestroy(comm) == 0

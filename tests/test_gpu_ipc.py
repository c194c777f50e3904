"""What the peer halo transport assumes of HIP IPC on this platform (csrc/csi_abi.hip peer_setup), checked between two
PROCESSES on one GPU: a sub-allocation of torch's caching allocator and a fine-grained hipExtMallocWithFlags buffer can be
exported as (hipIpcGetMemHandle of the allocation's base, offset) and opened, written and read by another process.  (The
cross-DEVICE part -- xGMI peer access -- needs two GPUs: tests/test_gpu_multirank.py.)"""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Handle(C.Structure):                 # hipIpcMemHandle_t: 64 opaque bytes, passed to hipIpcOpenMemHandle BY VALUE
    _fields_ = [("reserved", C.c_ubyte * 64)]


def _hip():
    L = C.CDLL("libamdhip64.so")
    L.hipGetErrorString.restype = C.c_char_p
    L.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    L.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
    return L


def _ck(L, rc, what):
    assert rc == 0, f"{what}: {L.hipGetErrorString(rc).decode()}"


def _child(conn):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = _hip()
    _ck(L, L.hipSetDevice(0), "hipSetDevice")
    out = {}
    for name, (handle, offset, n) in conn.recv().items():
        h = Handle.from_buffer_copy(handle)
        p = C.c_void_p()
        rc = L.hipIpcOpenMemHandle(C.byref(p), h, 1)             # hipIpcMemLazyEnablePeerAccess
        if rc != 0:
            out[name] = f"hipIpcOpenMemHandle: {L.hipGetErrorString(rc).decode()}"
            continue
        host = (C.c_double * n)()
        _ck(L, L.hipMemcpy(host, C.c_void_p(p.value + offset), n * 8, 2), "D2H")
        seen = np.frombuffer(host, dtype=np.float64).copy()
        reply = (C.c_double * n)(*[-x for x in seen])
        _ck(L, L.hipMemcpy(C.c_void_p(p.value + offset), reply, n * 8, 1), "H2D")
        _ck(L, L.hipDeviceSynchronize(), "sync")
        _ck(L, L.hipIpcCloseMemHandle(p), "close")
        out[name] = seen
    conn.send(out)


def test_ipc_of_torch_suballocation_and_finegrained_buffer():
    import torch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = _hip()
    n = 1000
    pad = torch.zeros(12345, device="cuda:0", dtype=torch.float64)        # so that the next block is not at its segment's base
    t = torch.arange(1, n + 1, device="cuda:0", dtype=torch.float64)
    torch.cuda.synchronize()
    exports, keep = {}, [pad]

    def export(name, ptr):
        base, size = C.c_void_p(), C.c_size_t()
        _ck(L, L.hipMemGetAddressRange(C.byref(base), C.byref(size), C.c_void_p(ptr)), "hipMemGetAddressRange")
        h = Handle()
        rc = L.hipIpcGetMemHandle(C.byref(h), base)
        assert rc == 0, f"hipIpcGetMemHandle({name}): {L.hipGetErrorString(rc).decode()}"
        exports[name] = (bytes(h), ptr - base.value, n)

    export("torch", t.data_ptr())
    fine = C.c_void_p()
    rc = L.hipExtMallocWithFlags(C.byref(fine), C.c_size_t(n * 8), 0x1)    # hipDeviceMallocFinegrained
    if rc == 0:
        src = (C.c_double * n)(*range(1, n + 1))
        _ck(L, L.hipMemcpy(fine, src, n * 8, 1), "H2D")
        export("finegrained", fine.value)
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    pr = ctx.Process(target=_child, args=(b,))
    pr.start()
    a.send(exports)
    got = a.recv()
    pr.join(timeout=120)
    assert pr.exitcode == 0
    want = np.arange(1, n + 1, dtype=np.float64)
    for name in exports:
        assert isinstance(got[name], np.ndarray), (name, got[name])
        assert np.array_equal(got[name], want), name                      # the other process saw this process's data ...
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), -want)                         # ... and this process sees what it wrote back
    if rc == 0:
        back = (C.c_double * n)()
        _ck(L, L.hipMemcpy(back, fine, n * 8, 2), "D2H")
        assert np.array_equal(np.frombuffer(back, dtype=np.float64), -want)
        L.hipFree(fine)

"""ctypes binding of the multi-GPU layer of libmsstitch.so (include/ms_dist.h) for tests and bench.py.

One `Dist` per rank (process or thread).  RCCL over xGMI when every rank drives its own GPU, the shared-memory host mailbox when ranks share a
device or there is none (protocol tests on CPU).  torch tensors stand in for device buffers; numpy arrays / bytes for host memory.
"""
import ctypes as C

import msstitch as ms

AUTO, RCCL, HOST = 0, 1, 2
MEM_DEVICE, MEM_HOST = 0, 1
ID_BYTES = 144
MAX_RANKS = 16

EXPORTS = [
    "ms_dist_unique_id", "ms_dist_create", "ms_dist_destroy", "ms_dist_get_info", "ms_dist_send", "ms_dist_recv", "ms_dist_group_begin",
    "ms_dist_group_end", "ms_dist_broadcast", "ms_dist_barrier", "ms_dist_gather_slabs", "ms_dist_mesh_exchange", "ms_dist_apply_meshes",
    "ms_dist_set_rccl_library", "ms_dist_rccl_library_path",
]


class Info(C.Structure):
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("transport", C.c_int), ("rccl_version", C.c_int), ("comm_nranks", C.c_int),
                ("device", C.c_int * MAX_RANKS), ("pci_bus_id", (C.c_char * 16) * MAX_RANKS)]


class MeshUpdate(C.Structure):
    _fields_ = [("swap_frame", C.c_longlong), ("version", C.c_int), ("n_views", C.c_int), ("rows", C.c_int), ("cols", C.c_int),
                ("mesh_x", C.POINTER(C.c_float)), ("mesh_y", C.POINTER(C.c_float))]


def set_rccl_library(path):
    """Name the RCCL library file (before the first RCCL id / communicator of the process); None restores the default search."""
    ms._chk(ms.load().ms_dist_set_rccl_library(None if path is None else str(path).encode()))


def rccl_library_path():
    """The file the product resolved RCCL's entry points from (realpath), or None when no RCCL loads."""
    buf = C.create_string_buffer(4096)
    if ms.load().ms_dist_rccl_library_path(buf, C.c_size_t(4096)) != 0:
        return None
    return buf.value.decode() or None


def unique_id(nranks, transport=AUTO):
    """Rank 0 makes the id; every rank passes the same bytes to Dist()."""
    buf = (C.c_ubyte * ID_BYTES)()
    ms._chk(ms.load().ms_dist_unique_id(int(transport), int(nranks), buf))
    return bytes(buf)


def id_transport(id_bytes):
    return int.from_bytes(id_bytes[128:132], "little")


def _ptr_bytes(buf):
    """(pointer, nbytes, memory kind) of a torch tensor (device or CPU) or a writable numpy array."""
    if hasattr(buf, "data_ptr"):
        assert buf.is_contiguous()
        return C.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size(), (MEM_DEVICE if buf.is_cuda else MEM_HOST)
    assert buf.flags["C_CONTIGUOUS"]
    return C.c_void_p(buf.ctypes.data), buf.nbytes, MEM_HOST


def _stream(buf=None):
    if buf is not None and hasattr(buf, "is_cuda") and buf.is_cuda:
        return ms._stream()
    try:
        import torch
        if torch.cuda.is_available():
            return ms._stream()
    except ImportError:
        pass
    return None


class Dist:
    def __init__(self, rank, nranks, id_bytes, device=0):
        self._h = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        lib = ms.load()
        lib.ms_dist_destroy.restype = None
        ms._chk(lib.ms_dist_create(C.byref(self._h), int(rank), int(nranks), id_bytes, int(device)))

    def close(self):
        if self._h:
            ms.load().ms_dist_destroy(self._h)
            self._h = C.c_void_p()

    def info(self):
        i = Info()
        ms._chk(ms.load().ms_dist_get_info(self._h, C.byref(i)))
        return {"rank": i.rank, "nranks": i.nranks, "transport": {RCCL: "rccl", HOST: "host"}.get(i.transport, "?"), "rccl_version": i.rccl_version,
                "comm_nranks": i.comm_nranks, "devices": [i.device[r] for r in range(i.nranks)],
                "pci_bus_ids": [bytes(i.pci_bus_id[r]).split(b"\0")[0].decode() for r in range(i.nranks)],
                "librccl_path": (rccl_library_path() if i.transport == RCCL else None)}

    def send(self, buf, peer):
        p, n, mem = _ptr_bytes(buf)
        ms._chk(ms.load().ms_dist_send(self._h, p, C.c_size_t(n), int(peer), mem, _stream(buf)))

    def recv(self, buf, peer):
        p, n, mem = _ptr_bytes(buf)
        ms._chk(ms.load().ms_dist_recv(self._h, p, C.c_size_t(n), int(peer), mem, _stream(buf)))

    def group_begin(self):
        ms._chk(ms.load().ms_dist_group_begin(self._h))

    def group_end(self):
        ms._chk(ms.load().ms_dist_group_end(self._h))

    def broadcast(self, buf, root):
        p, n, mem = _ptr_bytes(buf)
        ms._chk(ms.load().ms_dist_broadcast(self._h, p, C.c_size_t(n), int(root), mem, _stream(buf)))

    def barrier(self):
        ms._chk(ms.load().ms_dist_barrier(self._h, _stream()))

    def gather_slabs(self, slab, recv, sink=0):
        """slab: this rank's contiguous device tensor; recv: on the sink a list of nranks device tensors (entry `sink` may be None), else None."""
        p, n, _ = _ptr_bytes(slab)
        arr = None
        if self.rank == sink:
            arr = (C.c_void_p * self.nranks)(*[None if (t is None) else t.data_ptr() for t in recv])
        ms._chk(ms.load().ms_dist_gather_slabs(self._h, p, C.c_size_t(n), arr, int(sink), _stream(slab)))

    def mesh_exchange(self, root, update, n_views, rows, cols):
        """Collective.  update (root only, or None): (swap_frame, version, mesh_x, mesh_y) with float32 arrays (n_views, rows, cols).
        Returns None or (swap_frame, version, mesh_x, mesh_y) -- the same on every rank."""
        import numpy as np
        cap = n_views * rows * cols
        ox, oy = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        out = MeshUpdate(0, 0, 0, 0, 0, ox.ctypes.data_as(C.POINTER(C.c_float)), oy.ctypes.data_as(C.POINTER(C.c_float)))
        upd = None
        keep = None
        if update is not None and self.rank == root:
            sf, ver, mx, my = update
            mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
            keep = (mx, my)
            upd = C.byref(MeshUpdate(int(sf), int(ver), n_views, rows, cols, mx.ctypes.data_as(C.POINTER(C.c_float)), my.ctypes.data_as(C.POINTER(C.c_float))))
        have = C.c_int(0)
        ms._chk(ms.load().ms_dist_mesh_exchange(self._h, int(root), upd, C.byref(out), C.c_size_t(cap), C.byref(have), _stream()))
        del keep
        if not have.value:
            return None
        shp = (out.n_views, out.rows, out.cols)
        n = out.n_views * out.rows * out.cols
        return (out.swap_frame, out.version, ox[:n].reshape(shp).copy(), oy[:n].reshape(shp).copy())


def apply_meshes(comp, update, next_frame):
    """ms_set_mesh for every view once `next_frame` has reached the update's swap frame.  comp: msstitch.Compositor.  Returns True when applied."""
    import numpy as np
    sf, ver, mx, my = update
    mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
    u = MeshUpdate(int(sf), int(ver), mx.shape[0], mx.shape[1], mx.shape[2], mx.ctypes.data_as(C.POINTER(C.c_float)), my.ctypes.data_as(C.POINTER(C.c_float)))
    applied = C.c_int(0)
    ms._chk(ms.load().ms_dist_apply_meshes(comp._ctx, C.byref(u), C.c_longlong(int(next_frame)), C.byref(applied), ms._stream()))
    return bool(applied.value)

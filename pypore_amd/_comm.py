"""ctypes binding of libporeseg_comm.so (include/poreseg_comm.h): the multi-GPU entry points of the C ABI -- communicator
set-up and the boundary gather over RCCL.  No fallback: a missing library raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PORESEG_COMM_LIB") or os.path.join(_HERE, "libporeseg_comm.so")
ID_BYTES = 128
HEADER = 4
EXPORTS = ["ps_comm_unique_id", "ps_comm_init_rank", "ps_comm_init_all", "ps_comm_world", "ps_comm_rank", "ps_comm_destroy",
           "ps_gather_bounds", "ps_gather_bounds_all", "ps_comm_last_error"]
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("pypore_amd: %s not found -- build it with `make -C pypore_amd/csrc` (hipcc, RCCL)" % LIB_PATH)
    import torch  # noqa: F401  (first: one HIP runtime and one RCCL in the process, torch's)
    L = ctypes.CDLL(LIB_PATH)
    vp, P = ctypes.c_void_p, ctypes.POINTER
    L.ps_comm_unique_id.argtypes = [ctypes.c_char_p]
    L.ps_comm_init_rank.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, P(vp)]
    L.ps_comm_init_all.argtypes = [ctypes.c_int, P(ctypes.c_int), P(vp)]
    L.ps_comm_world.argtypes = [vp]
    L.ps_comm_rank.argtypes = [vp]
    L.ps_comm_destroy.argtypes = [vp]
    L.ps_comm_destroy.restype = None
    L.ps_gather_bounds.argtypes = [vp, vp, vp, ctypes.c_int64, vp]
    L.ps_gather_bounds_all.argtypes = [P(vp), ctypes.c_int, P(vp), P(vp), ctypes.c_int64, P(vp)]
    L.ps_comm_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError("poreseg_comm: %s (code %d)" % (lib().ps_comm_last_error().decode(), rc))


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    check(lib().ps_comm_unique_id(buf))
    return buf.raw


class Comm(object):
    """One rank of a communicator (ps_comm): Comm.for_rank(world, rank, id, device) in a process per GPU; Comm.all(n) in
    one process that drives n GPUs."""

    def __init__(self, handle):
        self.handle = handle
        self.world = lib().ps_comm_world(handle)
        self.rank = lib().ps_comm_rank(handle)

    @classmethod
    def for_rank(cls, world, rank, uid, device):
        h = ctypes.c_void_p()
        check(lib().ps_comm_init_rank(int(world), int(rank), uid, int(device), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def all(cls, ndev, devices=None):
        hs = (ctypes.c_void_p * ndev)()
        dv = (ctypes.c_int * ndev)(*devices) if devices is not None else None
        check(lib().ps_comm_init_all(int(ndev), dv, hs))
        return [cls(ctypes.c_void_p(h)) for h in hs]

    def gather_bounds(self, send, recv, stream=None):
        """send: int32 CUDA tensor [capacity] (element 0 = count, payload from HEADER on); recv: [world * capacity];
        asynchronous on `stream` (a torch.cuda.Stream; default: the current one)."""
        import torch
        assert send.is_cuda and recv.is_cuda and send.dtype == torch.int32 and recv.dtype == torch.int32
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == self.world * send.numel()
        st = stream if stream is not None else torch.cuda.current_stream(send.device)
        check(lib().ps_gather_bounds(self.handle, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                                     int(send.numel()), ctypes.c_void_p(st.cuda_stream)))

    def close(self):
        if self.handle:
            lib().ps_comm_destroy(self.handle)
            self.handle = None

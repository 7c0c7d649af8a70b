"""Process-wide device runtime: one HIP context (= one GPU, one video stream) per process.

The reference keeps one CUDA stream per TRTInference object (fastmot/utils/inference.py:67);
here every stage of one MOT instance shares a single fm_ctx so that the device-resident
track table, the per-frame embeddings and the frame buffers are visible to all stages
without host round trips.  Multi-GPU = one process per GPU (LOCAL_RANK selects the device),
which is also what the reference's process-global `Track._count` requires (track.py:130).
"""
import atexit
import os

from . import _lib

_CTX = None
SCRATCH_SLOT = 0   # reserved for the host-array KalmanFilter API (create/predict/update on ndarrays)


class SlotAllocator:
    """Hands out rows of the device track table (slot 0 is reserved)."""

    def __init__(self):
        self._free = []
        self._next = 1

    def alloc(self):
        if self._free:
            return self._free.pop()
        slot = self._next
        self._next += 1
        return slot

    def free(self, slot):
        if slot is not None and slot > 0:
            self._free.append(slot)

    @property
    def high_water(self):
        return self._next


def get_context():
    """Returns the process-wide context, creating it on the GPU picked by LOCAL_RANK
    (torchrun convention) or FASTMOT_DEVICE.  Raises if the library or a GPU is missing."""
    global _CTX
    if _CTX is None:
        n = _lib.device_count()
        if n <= 0:
            raise RuntimeError('no HIP device visible: fastmot_amd has no CPU fallback')
        device = int(os.environ.get('FASTMOT_DEVICE', os.environ.get('LOCAL_RANK', '0'))) % n
        _CTX = _lib.HipContext(device)
        _CTX.slots = SlotAllocator()
        _CTX.device_emb_host = None
        # destroy the context (streams, graphs, buffers) before the interpreter and the HIP runtime tear down;
        # registered after concurrent.futures' own hook was, so worker threads are joined first
        atexit.register(reset_context)
    return _CTX


def reset_context():
    global _CTX
    if _CTX is not None:
        _CTX.close()
    _CTX = None

"""Process-wide device runtime: one HIP context (= one GPU, one video stream) per process.

The reference keeps one CUDA stream per TRTInference object (fastmot/utils/inference.py:67);
here every stage of one MOT instance shares a single fm_ctx so that the device-resident
track table, the per-frame embeddings and the frame buffers are visible to all stages
without host round trips.  Multi-GPU = one process per GPU (LOCAL_RANK selects the device),
which is also what the reference's process-global `Track._count` requires (track.py:130).
"""
import atexit
import os
import platform

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


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if '-' in part:
            a, b = part.split('-')
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


def bind_to_gpu_numa_node(device):
    """One process per GPU: keep its threads -- and with them the page-locked frame buffers it allocates next (first
    touch) -- on the NUMA node the GPU hangs off.  The per-frame H2D copy (6.2 MB at 1080p) reads that memory; with the
    process floating over two sockets the copy took 0.2 ms longer on some boxes (BENCH_r03: 723 against 857 frames/s
    for the same command line, the frames-resident variant unaffected).  Best effort and Linux only: without sysfs
    topology, on a single-node host, or with FASTMOT_NUMA_BIND=0 nothing is changed.  Returns what was found."""
    info = {'gpu_numa_node': None, 'bound': False}
    if hasattr(os, 'sched_getaffinity'):
        info['affinity_before'] = sorted(os.sched_getaffinity(0))     # (what the embedding application had: bench.py hands
    try:                                                                 # it back to the CPU baseline's worker processes)
        bdf = _lib.device_pci_bus_id(device)
        info['pci'] = bdf
        node = int(open(f'/sys/bus/pci/devices/{bdf}/numa_node').read())
        info['gpu_numa_node'] = node
        nodes = [d for d in os.listdir('/sys/devices/system/node') if d.startswith('node') and d[4:].isdigit()]
        info['host_numa_nodes'] = len(nodes)
        if node < 0 or len(nodes) < 2:
            return info
        local = _cpulist(open(f'/sys/devices/system/node/node{node}/cpulist').read())
        allowed = os.sched_getaffinity(0)
        info['cpus_allowed'], info['cpus_local'] = len(allowed), len(allowed & local)
        if os.environ.get('FASTMOT_NUMA_BIND', '1') == '0' or not (allowed & local):
            return info
        os.sched_setaffinity(0, allowed & local)
        try:                                     # memory policy: prefer the node as well (set_mempolicy, MPOL_PREFERRED)
            if platform.machine() != 'x86_64':   # (raw syscall number 238 is x86-64's set_mempolicy only)
                raise OSError('set_mempolicy by number: x86-64 only')
            import ctypes
            mask = ctypes.c_ulong(1 << node)
            libc = ctypes.CDLL(None, use_errno=True)
            libc.syscall(ctypes.c_long(238), ctypes.c_int(1), ctypes.byref(mask), ctypes.c_ulong(8 * ctypes.sizeof(mask)))
        except (OSError, AttributeError, ValueError):
            pass
        info['bound'] = True
    except (OSError, ValueError, _lib.FastMOTHipError):
        pass
    return info


def numa_node_of(array):
    """NUMA node of the first page of a host array (move_pages query); None when it cannot be told."""
    try:
        if platform.machine() != 'x86_64':       # (raw syscall number 279 is x86-64's move_pages only)
            return None
        import ctypes
        libc = ctypes.CDLL(None, use_errno=True)
        page = ctypes.c_void_p(array.ctypes.data & ~4095)
        status = ctypes.c_int(-1)
        rc = libc.syscall(ctypes.c_long(279), ctypes.c_int(0), ctypes.c_ulong(1), ctypes.byref(page), None,
                          ctypes.byref(status), ctypes.c_int(0))
        return int(status.value) if rc == 0 and status.value >= 0 else None
    except (OSError, AttributeError, ValueError):
        return None


def get_context():
    """Returns the process-wide context, creating it on the GPU picked by LOCAL_RANK
    (torchrun convention) or FASTMOT_DEVICE.  Raises if the library or a GPU is missing."""
    global _CTX
    if _CTX is None:
        n = _lib.device_count()
        if n <= 0:
            raise RuntimeError('no HIP device visible: fastmot_amd has no CPU fallback')
        device = int(os.environ.get('FASTMOT_DEVICE', os.environ.get('LOCAL_RANK', '0'))) % n
        numa = bind_to_gpu_numa_node(device)     # before the context: its page-locked buffers and worker threads follow
        _CTX = _lib.HipContext(device)
        _CTX.numa = numa
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

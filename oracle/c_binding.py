"""ctypes access to the C restatement (oracle/ju_oracle_c.c).

Test infrastructure / CPU baseline only -- see the header of ju_oracle_c.c.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_DIR, "_build", "libju_oracle.so")
_LIB = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _DIR])
    return _PATH


def usable_cpus() -> int:
    """CPUs this process may really keep busy: the affinity mask, cut to the cgroup's CPU quota (v2 `cpu.max`, v1
    `cpu.cfs_quota_us`).  OpenMP's default is the number of logical CPUs it can see; on the GPU boxes that is 256 under a
    quota of 16, and the oversubscribed run is throttled to a quarter of the 16-thread speed
    (profiles/r04_cpu_scaling.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_PATH):
            build()
        lib = C.CDLL(_PATH)
        lib.juo_create.restype = C.c_void_p
        lib.juo_create.argtypes = [C.c_char_p, C.c_size_t]
        lib.juo_destroy.argtypes = [C.c_void_p]
        lib.juo_reset.argtypes = [C.c_void_p]
        lib.juo_run.restype = C.c_int
        lib.juo_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.juo_output_raw.restype = C.POINTER(C.c_float)
        lib.juo_output_raw.argtypes = [C.c_void_p]
        lib.juo_num_threads.restype = C.c_int
        lib.juo_set_num_threads.argtypes = [C.c_int]
        lib.juo_vector_bits.restype = C.c_int
        if "OMP_NUM_THREADS" not in os.environ:
            lib.juo_set_num_threads(usable_cpus())
        _LIB = lib
    return _LIB


class CSession:
    """Recurrent driver over the C restatement (same call shape as
    oracle.ju_oracle.Session)."""

    def __init__(self, blob: bytes, h: int, w: int):
        self.lib = load()
        self.h, self.w = h, w
        self.handle = self.lib.juo_create(blob, len(blob))
        if not self.handle:
            raise ValueError("ju_oracle_c rejected the model")

    def run(self, frame_bgrx: np.ndarray) -> np.ndarray:
        frame = np.ascontiguousarray(frame_bgrx, dtype=np.uint8)
        out = np.empty((4 * self.h, 4 * self.w, 4), np.uint8)
        self.lib.juo_run(self.handle, frame.ctypes.data, out.ctypes.data)
        return out

    def output_raw(self) -> np.ndarray:
        p = self.lib.juo_output_raw(self.handle)
        return np.ctypeslib.as_array(p, shape=(4 * self.h, 4 * self.w, 3)).copy()

    @property
    def threads(self) -> int:
        return self.lib.juo_num_threads()

    @property
    def vector_bits(self) -> int:
        """512 / 256: width of the convolution's blocks on this CPU; 0: the plain form (JUO_VECTOR_BITS=0)."""
        return self.lib.juo_vector_bits()

    def close(self):
        if self.handle:
            self.lib.juo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

"""ctypes binding of libJoshUpscale.so (the C ABI in include/joshupscale_amd.h).

Host-side mirror, in Python, of what the reference exposes to its callers:

* :class:`Runtime` <-> ``JoshUpscale::core::Runtime`` / ``createRuntime``
  (reference core/public/JoshUpscale/core.h:64-92): ``process_image`` is
  ``processImage`` on BGRX frames.
* :class:`Session` <-> the recurrent drivers of the reference's Python scripts
  (scripts/inference/onnx/inference.py:46-94,
  scripts/inference/tensorrt/inference.py:60-193): ``run(image)`` takes one
  ``[H, W, 3|4]`` uint8 BGR(X) frame and returns the upscaled frame, the state
  living inside the runtime.

There is NO CPU fallback: if the HIP library is missing or no GPU is visible the
constructors raise.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_LIB_NAME = "libJoshUpscale.so"
# The same objects with the test and measurement hooks of include/joshupscale_amd_test.h compiled in
# (ju_debug_*, ju_read_tensor, ju_time_steps).  The product library exports none of them.
_TEST_LIB_NAME = "libJoshUpscale_test.so"
_LIBS: dict = {}

LOC_CPU, LOC_DEVICE, LOC_GRAPHICS_RESOURCE = 0, 1, 2
DTYPE_DEFAULT, DTYPE_F16, DTYPE_BF16 = -1, 0, 1
# the 64->64 block convolutions of the generator on e4m3 operands (csrc/fp8.h), every
# other layer and the residual stream fp16
DTYPE_FP8 = 2
DTYPE_NAMES = {DTYPE_F16: "fp16", DTYPE_BF16: "bf16", DTYPE_FP8: "fp8"}


class JuImage(C.Structure):
    """``ju_image`` == ``JoshUpscale::core::Image`` (core.h:32-38)."""
    _fields_ = [("ptr", C.c_void_p), ("location", C.c_uint8),
                ("stride", C.c_ssize_t), ("width", C.c_size_t),
                ("height", C.c_size_t)]


LOG_CALLBACK = C.CFUNCTYPE(None, C.c_char_p, C.c_int, C.c_char_p, C.c_void_p)


class JoshUpscaleError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


def hooks_default() -> bool:
    """``JU_TEST_HOOKS=1`` (set by tests/conftest.py and the developer tools): the process works
    through libJoshUpscale_test.so.  Unset -- every caller of the product -- it is the product library."""
    return os.environ.get("JU_TEST_HOOKS", "0") == "1"


def library_path(hooks: Optional[bool] = None) -> str:
    """In-tree HIP library (``hooks``: its test flavour); ``JU_LIBRARY`` points at another build
    (A/B timing: such a build carries the hooks or not as it was linked)."""
    override = os.environ.get("JU_LIBRARY")
    if override:
        return override
    hooks = hooks_default() if hooks is None else hooks
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                        _TEST_LIB_NAME if hooks else _LIB_NAME)


_P = C.POINTER
_PRODUCT_SIGS = {
    "ju_create": (C.c_int, [C.c_int, C.c_char_p, _P(C.c_void_p)]),
    "ju_create_from_memory": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, _P(C.c_void_p)]),
    "ju_validate_model": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ju_destroy": (None, [C.c_void_p]),
    "ju_process": (C.c_int, [C.c_void_p, _P(JuImage), _P(JuImage)]),
    "ju_process_batch": (C.c_int, [C.c_void_p, _P(JuImage), _P(JuImage), C.c_int]),
    "ju_prepare_batch": (C.c_int, [C.c_void_p, _P(JuImage), _P(JuImage), C.c_int, _P(C.c_int)]),
    "ju_set_lookahead": (C.c_int, [C.c_void_p, C.c_int]),
    "ju_enqueue": (C.c_int, [C.c_void_p, _P(JuImage), _P(JuImage)]),
    "ju_synchronize": (C.c_int, [C.c_void_p]),
    "ju_prepare_frames": (C.c_int, [C.c_void_p, _P(JuImage), _P(JuImage), _P(C.c_int)]),
    "ju_get_size": (C.c_int, [C.c_void_p] + [_P(C.c_size_t)] * 4),
    "ju_reset": (C.c_int, [C.c_void_p]),
    "ju_last_error": (C.c_char_p, []),
    "ju_set_log_callback": (None, [LOG_CALLBACK, C.c_void_p]),
    "ju_get_gl_device_index": (C.c_int, [_P(C.c_int)]),
    "ju_get_gl_image": (C.c_int, [C.c_uint32, C.c_int, _P(JuImage)]),
    "ju_release_gl_image": (None, [_P(JuImage)]),
    "ju_get_dtype": (C.c_int, [C.c_void_p]),
    "ju_get_stat": (C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_double)]),
    "ju_version": (C.c_char_p, []),
    "ju_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ju_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_void_p)]),
    "ju_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "ju_comm_allreduce_max": (C.c_int, [C.c_void_p, _P(C.c_double)]),
    "ju_comm_count": (C.c_int, [C.c_void_p, _P(C.c_int)]),
    "ju_comm_destroy": (None, [C.c_void_p]),
}
# include/joshupscale_amd_test.h
_HOOK_SIGS = {
    "ju_debug_fake_gl_texture": (C.c_int, [C.c_uint32, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "ju_debug_fake_gl_counters": (None, [_P(C.c_int)] * 4),
    "ju_debug_e4m3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ju_read_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "ju_time_steps": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _P(C.c_double), _P(C.c_int), _P(C.c_double)]),
    "ju_debug_set": (C.c_int, [C.c_char_p, C.c_int]),
}
PRODUCT_SYMBOLS = tuple(sorted(_PRODUCT_SIGS))
HOOK_SYMBOLS = tuple(sorted(_HOOK_SIGS))


def load_library(hooks: Optional[bool] = None) -> C.CDLL:
    """Load the HIP library; fails loudly (no fallback) when it is not built.  ``hooks`` selects the
    test flavour (default: ``JU_TEST_HOOKS``); a library that lacks a hook raises when the hook is used."""
    hooks = hooks_default() if hooks is None else hooks
    path = library_path(hooks)
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). "
            "joshupscale_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _PRODUCT_SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _HOOK_SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if hooks:
                raise ImportError(f"{path} does not export {name}: not a -DJU_TEST_HOOKS build") from None
            continue
        fn.restype = res
        fn.argtypes = args
    _LIBS[path] = lib
    return lib


def _hook(lib: C.CDLL, name: str):
    try:
        return getattr(lib, name)
    except AttributeError:
        raise RuntimeError(f"{name} is a test hook (include/joshupscale_amd_test.h): the product library does not "
                           "export it; set JU_TEST_HOOKS=1 or pass hooks=True to work through "
                           "libJoshUpscale_test.so") from None


def _check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise JoshUpscaleError(rc, lib.ju_last_error().decode(errors="replace"))


def validate_model(model: bytes) -> None:
    """Check a container with the C++ loader (``ju_validate_model``): no GPU needed.
    Raises :class:`JoshUpscaleError` with the loader's message."""
    lib = load_library()
    buf = bytes(model)
    _check(lib, lib.ju_validate_model(buf, len(buf)))


class Runtime:
    """One recurrent SR stream on one GPU (``JoshUpscale::core::Runtime``)."""

    def __init__(self, model, device: int = 0, dtype: int = DTYPE_DEFAULT, hooks: Optional[bool] = None):
        """``model``: path of a .jupw file, or its bytes.  ``hooks``: through libJoshUpscale_test.so
        (``read_tensor`` / ``time_steps`` need it; default: ``JU_TEST_HOOKS``)."""
        self._lib = load_library(hooks)
        self._h = C.c_void_p()
        if isinstance(model, (bytes, bytearray, memoryview)):
            buf = bytes(model)
            _check(self._lib, self._lib.ju_create_from_memory(
                device, buf, len(buf), dtype, C.byref(self._h)))
        else:
            if dtype != DTYPE_DEFAULT:
                with open(model, "rb") as f:
                    buf = f.read()
                _check(self._lib, self._lib.ju_create_from_memory(
                    device, buf, len(buf), dtype, C.byref(self._h)))
            else:
                _check(self._lib, self._lib.ju_create(
                    device, os.fsencode(model), C.byref(self._h)))
        w = [C.c_size_t() for _ in range(4)]
        _check(self._lib, self._lib.ju_get_size(self._h, *[C.byref(x) for x in w]))
        self.input_width, self.input_height, self.output_width, self.output_height = (
            x.value for x in w)
        self.device = device

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ju_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the boundary ------------------------------------------------------
    @property
    def dtype(self) -> int:
        return self._lib.ju_get_dtype(self._h)

    def process(self, inp: JuImage, out: JuImage) -> None:
        """Raw ``processImage`` on two image descriptors (synchronous)."""
        _check(self._lib, self._lib.ju_process(self._h, C.byref(inp), C.byref(out)))

    def process_batch(self, inputs, outputs) -> None:
        """``ju_process_batch``: consecutive frames of the stream in one synchronous call (frame look-ahead --
        every input must hold its pixels now); the bytes of ``process`` called frame by frame."""
        n = len(inputs)
        if len(outputs) != n:
            raise ValueError("as many outputs as inputs")
        ins, outs = (JuImage * n)(*inputs), (JuImage * n)(*outputs)
        _check(self._lib, self._lib.ju_process_batch(self._h, ins, outs, n))

    def prepare_batch(self, inputs, outputs) -> int:
        """``ju_prepare_batch``: capture the graphs of a tuple of device frame buffers that will go through
        ``process_batch`` as one pass; returns the graphs captured (0: the tuple will not run as one pass)."""
        n = len(inputs)
        if len(outputs) != n:
            raise ValueError("as many outputs as inputs")
        ins, outs = (JuImage * n)(*inputs), (JuImage * n)(*outputs)
        got = C.c_int()
        _check(self._lib, self._lib.ju_prepare_batch(self._h, ins, outs, n, C.byref(got)))
        return got.value

    def set_lookahead(self, frames: int) -> None:
        """``ju_set_lookahead``: frames per look-ahead pass of ``process_batch`` (1 = frame by frame, at most 8)."""
        _check(self._lib, self._lib.ju_set_lookahead(self._h, int(frames)))

    def enqueue(self, inp: JuImage, out: JuImage) -> None:
        _check(self._lib, self._lib.ju_enqueue(self._h, C.byref(inp), C.byref(out)))

    def synchronize(self) -> None:
        _check(self._lib, self._lib.ju_synchronize(self._h))

    def prepare_frames(self, inp: JuImage, out: JuImage) -> int:
        """``ju_prepare_frames``: capture the graphs of a device frame-buffer pair now
        (the reference captures in its constructor); returns the graphs captured."""
        n = C.c_int()
        _check(self._lib, self._lib.ju_prepare_frames(self._h, C.byref(inp), C.byref(out), C.byref(n)))
        return n.value

    def reset(self) -> None:
        _check(self._lib, self._lib.ju_reset(self._h))

    def process_image(self, frame_bgrx: np.ndarray,
                      out: Optional[np.ndarray] = None) -> np.ndarray:
        """Host frames: ``[H, W, 4]`` uint8 in, ``[4H, 4W, 4]`` uint8 out.  Any
        row stride (also negative, i.e. a ``[::-1]`` view) is passed through."""
        if frame_bgrx.dtype != np.uint8 or frame_bgrx.ndim != 3 or frame_bgrx.shape[2] != 4:
            raise ValueError("expected a [H, W, 4] uint8 BGRX frame")
        if frame_bgrx.strides[2] != 1 or frame_bgrx.strides[1] != 4:
            frame_bgrx = np.ascontiguousarray(frame_bgrx)
        if out is None:
            out = np.empty((self.output_height, self.output_width, 4), np.uint8)
        if out.strides[2] != 1 or out.strides[1] != 4:
            raise ValueError("output must have contiguous pixels")
        self.process(host_image(frame_bgrx), host_image(out))
        return out

    def device_image(self, ptr: int, width: int, height: int,
                     stride: Optional[int] = None) -> JuImage:
        return JuImage(ptr, LOC_DEVICE, width * 4 if stride is None else stride,
                       width, height)

    # -- introspection -------------------------------------------------------
    def read_tensor(self, name: str) -> np.ndarray:
        n = C.c_size_t()
        read = _hook(self._lib, "ju_read_tensor")
        _check(self._lib, read(self._h, name.encode(), None, 0, C.byref(n)))
        arr = np.empty(n.value, np.float32)
        _check(self._lib, read(self._h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.size, C.byref(n)))
        return arr

    def stat(self, key: str) -> float:
        """``ju_get_stat``: "graph_replays", "eager_runs", "direct_graphs",
        "resident_tower", "resident_flow", "launches_per_frame"."""
        v = C.c_double()
        _check(self._lib, self._lib.ju_get_stat(self._h, key.encode(), C.byref(v)))
        return v.value

    def time_steps(self, tag: str, iters: int) -> Tuple[float, int, float]:
        """(ms per launch, launches per repetition, FLOPs per repetition)."""
        ms, n, fl = C.c_double(), C.c_int(), C.c_double()
        _check(self._lib, _hook(self._lib, "ju_time_steps")(
            self._h, tag.encode(), iters, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """``ju_comm_unique_id``: the RCCL communicator id rank 0 hands to the other ranks."""
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib, lib.ju_comm_unique_id(buf))
    return buf.raw


class Comm:
    """``ju_comm``: the C layer's RCCL communicator (one rank per GPU).  Its one job is the
    start-up broadcast of the model container (BASELINE.json config 4)."""

    def __init__(self, unique_id: bytes, rank: int, world_size: int, device: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be COMM_ID_BYTES long")
        self._lib = load_library()
        self._h = C.c_void_p()
        self.rank, self.world_size = rank, world_size
        _check(self._lib, self._lib.ju_comm_create(unique_id, rank, world_size, device, C.byref(self._h)))

    def broadcast(self, blob: Optional[bytes], size: int, root: int = 0) -> bytes:
        """Every rank passes ``size``; ``root`` also the bytes.  Returns the bytes."""
        buf = C.create_string_buffer(bytes(blob), size) if self.rank == root else C.create_string_buffer(size)
        _check(self._lib, self._lib.ju_comm_broadcast(self._h, buf, size, root))
        return buf.raw

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(value)
        _check(self._lib, self._lib.ju_comm_allreduce_max(self._h, C.byref(v)))
        return v.value

    def count(self) -> int:
        n = C.c_int()
        _check(self._lib, self._lib.ju_comm_count(self._h, C.byref(n)))
        return n.value

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ju_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


def gl_image(texture: int, output: bool) -> JuImage:
    """``ju_get_gl_image`` (``getGLImage``, reference core.h:61-62): a registered OpenGL
    texture as a GRAPHICS_RESOURCE image.  Release with :func:`release_gl_image`."""
    lib = load_library()
    img = JuImage()
    _check(lib, lib.ju_get_gl_image(texture, 1 if output else 0, C.byref(img)))
    return img


def release_gl_image(img: JuImage) -> None:
    load_library().ju_release_gl_image(C.byref(img))


def host_image(arr: np.ndarray) -> JuImage:
    """Describe a ``[H, W, 4]`` uint8 numpy array (any row stride) as an image."""
    return JuImage(arr.ctypes.data, LOC_CPU, arr.strides[0], arr.shape[1], arr.shape[0])


class Session:
    """Recurrent driver with the reference scripts' call shape
    (scripts/inference/onnx/inference.py:46-94): ``Session(model).run(image)``.
    Accepts ``[H, W, 3]`` BGR (as ``cv2.imread`` yields) or ``[H, W, 4]`` BGRX and
    returns the same number of channels."""

    def __init__(self, model, device: int = 0, dtype: int = DTYPE_DEFAULT, hooks: Optional[bool] = None):
        self.runtime = Runtime(model, device, dtype, hooks)

    def run(self, image: np.ndarray) -> np.ndarray:
        if image.ndim == 4 and image.shape[0] == 1:
            image = image[0]
        ch = image.shape[2]
        if ch == 3:
            frame = np.empty(image.shape[:2] + (4,), np.uint8)
            frame[..., :3] = image
            frame[..., 3] = 255
        else:
            frame = image
        out = self.runtime.process_image(np.ascontiguousarray(frame, dtype=np.uint8))
        return out[..., :3].copy() if ch == 3 else out

    def reset(self) -> None:
        self.runtime.reset()

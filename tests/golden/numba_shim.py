"""IN-CONTAINER ONLY: a CPU interpreter for the subset of ``numba`` / ``numba.cuda`` that the reference's
``evaluate/rotate_iou.py`` (K18) and ``evaluate/eval2.py`` use, so that the reference's OWN Python runs here
unmodified (imported read-only from /root/reference) and produces the committed fixtures
``g9_rotate_iou_ref.npz`` and ``g10_ap_eval_ref.npz``.  Nothing of the reference is copied: this file only
supplies the runtime the reference imports.

What is emulated, and how faithfully:

* ``numba.jit`` / ``cuda.jit(..., device=True)``  -> identity decorators (the function body runs as Python);
* ``cuda.jit(sig)`` without ``device=True``       -> a launcher: ``kernel[grid, block, stream](*args)`` runs the
  body once per thread with ``cuda.blockIdx`` / ``cuda.threadIdx`` set.  ``cuda.syncthreads()`` is honoured by
  running every block in phases: all threads run up to barrier k (a private exception unwinds them), then all
  run up to barrier k + 1 with the first k barriers as no-ops -- correct for kernels whose pre-barrier code is
  idempotent (rotate_iou.py:261-291 only copies boxes into shared memory before its single barrier);
  ``cuda.shared.array`` returns the SAME storage to every thread of a block in every phase;
* ``cuda.local.array`` / device arrays            -> ``F32Array``; **numba's typing rules** are reproduced by the
  scalar class ``F32``: float32 (op) float32 -> float32 with one rounding; float32 (op) int or Python float ->
  float64 (numba types literals as int64 / float64 and promotes); a store into a float32 array rounds to
  float32; ``math.sqrt/cos/sin`` of a float32 return float32 (sqrt correctly rounded, cos/sin :=
  (float) f64 libm -- the arithmetic contract of DESIGN.md section 3; CUDA's ``__nv_cosf`` is not available).
  Plain numpy float32 scalars would NOT do: under NEP 50 ``np.float32(x) / 2.0`` stays float32, whereas numba
  makes it float64 (rotate_iou.py:19-20,24-29: the triangle areas and their sum are float64);
* the reference's intersection buffer holds 8 points (rotate_iou.py:234) but ``quadrilateral_intersection`` can
  emit more (identical / edge-sharing boxes): on CUDA that is an out-of-bounds local-memory write = undefined
  behaviour.  Local arrays here carry slack so the evaluation completes, and every pair that touched the slack
  is reported through ``overflow_mask`` -- fixtures mark those pairs "undefined in the reference".
"""
import math as _math
import sys
import types

import numpy as np

_f32 = np.float32
_SLACK = 64


class F32:
    """A float32 value with numba's promotion rules (see module docstring)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = _f32(v)

    # float32 (op) float32 -> float32; anything else -> float64 (Python float)
    def __add__(self, o):
        return F32(self.v + o.v) if type(o) is F32 else float(self.v) + o

    def __radd__(self, o):
        return o + float(self.v)

    def __sub__(self, o):
        return F32(self.v - o.v) if type(o) is F32 else float(self.v) - o

    def __rsub__(self, o):
        return o - float(self.v)

    def __mul__(self, o):
        return F32(self.v * o.v) if type(o) is F32 else float(self.v) * o

    def __rmul__(self, o):
        return o * float(self.v)

    def __truediv__(self, o):
        if type(o) is F32:
            return F32(self.v / o.v)
        return _div64(float(self.v), o)

    def __rtruediv__(self, o):
        return _div64(o, float(self.v))

    def __neg__(self):
        return F32(-self.v)

    def __abs__(self):
        return F32(abs(self.v))

    def __float__(self):
        return float(self.v)

    def _c(self, o):
        return float(o.v) if type(o) is F32 else o

    def __lt__(self, o):
        return float(self.v) < self._c(o)

    def __le__(self, o):
        return float(self.v) <= self._c(o)

    def __gt__(self, o):
        return float(self.v) > self._c(o)

    def __ge__(self, o):
        return float(self.v) >= self._c(o)

    def __eq__(self, o):
        return float(self.v) == self._c(o)

    def __ne__(self, o):
        return float(self.v) != self._c(o)

    __hash__ = None

    def __repr__(self):
        return "F32(%r)" % float(self.v)


def _div64(a, b):
    try:
        return a / b
    except ZeroDivisionError:                      # IEEE result, like the device
        return float(np.float64(a) / np.float64(b))


def _to32(val):
    return val.v if type(val) is F32 else _f32(val)


class _State:
    overflow = False


class F32Array:
    """1-D float32 storage; slices are views; ``n`` = declared length (indices beyond it hit the slack)."""
    __slots__ = ("a", "n")

    def __init__(self, a, n=None):
        self.a = a
        self.n = len(a) if n is None else n

    def __getitem__(self, k):
        if type(k) is slice:
            return F32Array(self.a[k])
        if k >= self.n:
            _State.overflow = True
        return F32(self.a[k])

    def __setitem__(self, k, val):
        if type(k) is slice:
            self.a[k] = _to32(val)
            return
        if k >= self.n:
            _State.overflow = True
        self.a[k] = _to32(val)

    def __len__(self):
        return self.n

    def reshape(self, *a):
        return self

    def copy_to_host(self, dst=None, stream=None):
        if dst is None:
            return self.a.copy()
        dst[...] = self.a.reshape(dst.shape)
        return dst


class _OutArray(F32Array):
    """The kernel's result array: the store of a pair's value closes that pair's evaluation, so the overflow
    flag raised since the previous store belongs to this element."""
    __slots__ = ("undefined",)

    def __init__(self, a):
        super().__init__(a)
        self.undefined = np.zeros(len(a), dtype=bool)

    def __setitem__(self, k, val):
        if _State.overflow:
            self.undefined[k] = True
            _State.overflow = False
        self.a[k] = _to32(val)


class _Math:
    """``math`` as numba types it for float32 arguments (installed as the reference module's ``math``)."""

    def __getattr__(self, name):
        return getattr(_math, name)

    @staticmethod
    def sqrt(x):
        if type(x) is F32:
            return F32(np.sqrt(x.v))
        return _math.sqrt(x)

    @staticmethod
    def cos(x):
        if type(x) is F32:
            return F32(_math.cos(float(x.v)))
        return _math.cos(x)

    @staticmethod
    def sin(x):
        if type(x) is F32:
            return F32(_math.sin(float(x.v)))
        return _math.sin(x)


class _Barrier(Exception):
    pass


class _Idx:
    x = y = z = 0


class _Cuda(types.ModuleType):
    def __init__(self):
        super().__init__("numba.cuda")
        self.blockIdx, self.threadIdx, self.blockDim, self.gridDim = _Idx(), _Idx(), _Idx(), _Idx()
        self._shared, self._shared_i = [], 0
        self._barrier_i, self._barrier_stop = 0, 0
        self.last_undefined = None
        cu = self

        class _Local:
            @staticmethod
            def array(shape, dtype=None):
                n = int(shape[0]) if isinstance(shape, (tuple, list)) else int(shape)
                return F32Array(np.zeros(n + _SLACK, dtype=_f32), n)

        class _Shared:
            @staticmethod
            def array(shape, dtype=None):
                n = int(shape[0]) if isinstance(shape, (tuple, list)) else int(shape)
                if cu._shared_i == len(cu._shared):
                    cu._shared.append(F32Array(np.zeros(n, dtype=_f32)))
                arr = cu._shared[cu._shared_i]
                cu._shared_i += 1
                return arr

        self.local, self.shared = _Local, _Shared

    # ---- decorators
    def jit(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        if k.get("device"):
            return lambda f: f
        return lambda f: _Kernel(self, f)

    def syncthreads(self):
        self._barrier_i += 1
        if self._barrier_i > self._barrier_stop:
            raise _Barrier()

    # ---- host API used by rotate_iou.py:294-329
    def select_device(self, i):
        return None

    def stream(self):
        return _Stream()

    def to_device(self, arr, stream=None):
        assert arr.dtype == np.float32 and arr.ndim == 1
        return _OutArray(np.array(arr, dtype=_f32, copy=True))


class _Stream:
    def auto_synchronize(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Kernel:
    def __init__(self, cu, fn):
        self.cu, self.fn = cu, fn

    def __getitem__(self, cfg):
        grid, block = cfg[0], cfg[1]
        grid = tuple(grid) if isinstance(grid, (tuple, list)) else (grid,)
        grid = grid + (1,) * (3 - len(grid))
        block = int(block[0] if isinstance(block, (tuple, list)) else block)
        cu, fn = self.cu, self.fn

        def launch(*args):
            _State.overflow = False
            with np.errstate(all="ignore"):
                for bx in range(int(grid[0])):
                    for by in range(int(grid[1])):
                        cu.blockIdx.x, cu.blockIdx.y = bx, by
                        cu.blockDim.x = block
                        cu._shared = []
                        stop = 0
                        while True:                       # phase `stop`: run every thread up to barrier stop + 1
                            hit = False
                            cu._barrier_stop = stop
                            for tx in range(block):
                                cu.threadIdx.x = tx
                                cu._shared_i, cu._barrier_i = 0, 0
                                try:
                                    fn(*args)
                                except _Barrier:
                                    hit = True
                            if not hit:
                                break
                            stop += 1
            outs = [a for a in args if isinstance(a, _OutArray)]
            cu.last_undefined = outs[-1].undefined if outs else None
        return launch


def install():
    """Put the shim into ``sys.modules`` as ``numba`` / ``numba.cuda``.  Returns the cuda shim."""
    numba = types.ModuleType("numba")
    cu = _Cuda()

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    numba.jit = jit
    numba.float32 = np.float32
    numba.int32 = np.int32
    numba.cuda = cu
    sys.modules["numba"] = numba
    sys.modules["numba.cuda"] = cu
    return cu


def import_reference_rotate_iou():
    """The reference's evaluate/rotate_iou.py running on this interpreter.  Returns (module, cuda shim)."""
    import importlib
    cu = install()
    if "/root/reference/evaluate" not in sys.path:
        sys.path.append("/root/reference/evaluate")
    sys.modules.pop("rotate_iou", None)
    mod = importlib.import_module("rotate_iou")
    assert mod.__file__.startswith("/root/reference/"), mod.__file__
    mod.math = _Math()                               # numba's float32 overloads of sqrt / cos / sin
    return mod, cu


def reference_rotate_iou_eval(mod, cu, boxes, query_boxes, criterion=-1):
    """``rotate_iou_gpu_eval`` of the reference (host function + kernel, rotate_iou.py:261-329), plus the mask of
    pairs whose evaluation overran the reference's 8-point intersection buffer."""
    out = mod.rotate_iou_gpu_eval(np.asarray(boxes), np.asarray(query_boxes), criterion)
    und = cu.last_undefined
    und = np.zeros(out.shape, bool) if und is None or out.size == 0 else und.reshape(out.shape).copy()
    return out, und

"""Launch tape: record the kernel launches of one training step, replay them from C.

At the reference's per-GPU batch of 16 (scripts_psld/sota/uncond/cifar10/train_uncond_psld.sh:25-30) the step
is ~2700 launches and the Python executor needs longer to issue them (~11 us each) than the GPU needs to run them.  A
hipGraph of the step removes the host cost but serialises: a cross-stream edge costs 3.5 us inside a graph and the
weight-gradient side stream overlaps almost nothing there (profiles/r02/README.md).  The tape keeps the launches ordinary:

* recording happens while ``SDEWrapper`` runs the step once under stream capture (wrapper.py ``_graphed_step``), so every
  buffer the step allocates comes from the capture's private pool and keeps its address for as long as the graph object
  lives; the graph itself is never replayed;
* every ``libpsld_hip`` launch is noted as (function index, argument words) by a proxy that stands in for the loaded
  library (``_lib.load()`` hands it out while a recording is active); ctypes structures passed by reference are copied;
* the forks / joins between the compute stream and the side stream (score_fn.py ``flush_side`` / ``join_side``) become
  ``PSLD_TAPE_EDGE`` entries;
* the few ATen kernels PyTorch itself launches inside the step (the autograd glue of ``loss.backward()``, the time
  rescaling; ``tools/count_torch_ops.py``) are seen by a ``TorchDispatchMode`` and replayed as Python callables that
  write into the recorded output tensors; they split the tape into segments.

``replay()`` then costs one ``psld_tape_replay`` call per segment (include/psld_hip.h), which walks the entries in C.

Recording is process-wide (the proxy replaces what ``_lib.load()`` returns for every caller): nothing else may launch
through the library from another thread while a tape records.  Host memory a taped call points into - by-reference
structures (copied here) and the FIR taps of ``psld_upfirdn2d_f32`` (``_lib.keep_host_memory``) - lives as long as the tape.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Callable, List

import numpy as np
import torch
import torch.utils._python_dispatch as _pd
from torch.utils._pytree import tree_flatten

from . import _lib

MAX_ARGS = 24
ENTRY = np.dtype([("fn", "<i4"), ("nargs", "<i4"), ("a", "<u8", (MAX_ARGS,))])     # struct psld_tape_entry
EDGE = -1
_M64 = (1 << 64) - 1

# ATen ops that launch nothing: allocation, views, bookkeeping
_NO_KERNEL = {"aten::" + n for n in (
    "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "detach", "alias", "view", "_unsafe_view",
    "reshape", "_reshape_alias", "as_strided", "slice", "select", "permute", "transpose", "t", "unsqueeze", "squeeze",
    "expand", "split", "split_with_sizes", "chunk", "unbind", "narrow", "record_stream", "is_pinned", "lift_fresh",
    "view_as", "expand_as", "unfold", "diagonal", "movedim", "swapaxes", "flatten", "unflatten", "is_same_size",
    "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "stride", "size", "numel", "dim", "is_contiguous")}


def _f32_bits(v) -> int:
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


def _f64_bits(v) -> int:
    return struct.unpack("<Q", struct.pack("<d", float(v)))[0]


class LaunchTape:
    def __init__(self):
        self.lib = _lib.load_real()
        self.segments: List[tuple] = []       # ("c", entries ndarray, its address, count) | ("py", callable, name)
        self._cur: List[tuple] = []
        self._keep: List[object] = []         # structure copies, tensors of the ATen callables
        self._events: List[int] = []
        self.n_launches = self.n_edges = 0
        self.aten_ops: List[str] = []
        self.aten_outputs: List[tuple] = []   # (label, output tensors) of the functional ATen ops (debugging aid)
        self.stream = None                    # torch stream the step was recorded on (its compute stream)
        self._streams: set = set()            # that stream and every stream an edge has forked to / joined from
        self._failed = C.c_int(-1)
        self.through_stubs = False            # tests: execute every recorded launch through its C stub while recording

    # -- recording ---------------------------------------------------------------------------------------------------
    def _convert(self, name, args):
        _, argtypes = _lib.SIGNATURES[name]
        if len(args) != len(argtypes):
            raise TypeError(f"{name}: {len(args)} arguments, the signature has {len(argtypes)}")
        words = []
        for v, t in zip(args, argtypes):
            if t is C.c_int:
                words.append(int(v) & 0xFFFFFFFF)
            elif t is C.c_longlong or t is C.c_ulonglong:
                words.append(int(v) & _M64)
            elif t is C.c_float:
                words.append(_f32_bits(v))
            elif t is C.c_double:
                words.append(_f64_bits(v))
            else:                                               # c_void_p or POINTER(struct)
                words.append(self._pointer_word(v))
        return words

    def _pointer_word(self, v) -> int:
        if v is None:
            return 0
        if isinstance(v, int):
            return v & _M64
        obj = getattr(v, "_obj", None)                          # ctypes.byref(structure)
        if obj is not None:
            v = obj
        if isinstance(v, C.Structure):
            cp = type(v).from_buffer_copy(v)                    # the caller may reuse / mutate its structure
            self._keep.append(cp)
            return C.addressof(cp)
        if isinstance(v, C.c_void_p):
            return int(v.value or 0)
        raise TypeError(f"launch tape: cannot record a pointer argument of type {type(v)}")

    def add_launch(self, fn_index: int, name: str, args):
        words = self._convert(name, args)
        # Recording is process-wide (module docstring): a launch that some OTHER thread issues through the library while the
        # step records would be replayed on every step.  The step itself launches only on its compute stream and on the
        # streams it has forked to (every fork is noted as an edge BEFORE the launches behind it): anything else is refused.
        if self.stream is not None and not self.through_stubs:      # (the stub self-test issues a whole EAGER step, any stream)
            allowed = self._streams or {int(self.stream.cuda_stream)}
            if words[-1] not in allowed:
                raise RuntimeError(f"launch tape: {name} launched on stream {words[-1]:#x}, which is neither the recorded compute "
                                   "stream nor a stream forked from it - another thread launching during the recording?")
        self._cur.append((fn_index, words))
        self.n_launches += 1

    def add_edge(self, src_stream: int, dst_stream: int):
        if self.stream is not None:
            if not self._streams:
                self._streams.add(int(self.stream.cuda_stream))
            self._streams.update((int(src_stream), int(dst_stream)))
        ev = self.lib.psld_tape_event_create()
        if not ev:
            raise RuntimeError("launch tape: hipEventCreate failed")
        self._events.append(ev)
        self._cur.append((EDGE, [src_stream, dst_stream, ev]))
        self.n_edges += 1

    def add_callable(self, fn: Callable[[], None], name: str, keep=()):
        self._flush()
        self.segments.append(("py", fn, name))
        self.aten_ops.append(name)
        self._keep.extend(keep)

    def _flush(self):
        if not self._cur:
            return
        arr = np.zeros(len(self._cur), dtype=ENTRY)
        for i, (fn, words) in enumerate(self._cur):
            arr["fn"][i] = fn
            arr["nargs"][i] = len(words)
            arr["a"][i, :len(words)] = np.array(words, dtype=np.uint64)
        self.segments.append(("c", arr, arr.ctypes.data, len(arr)))
        self._cur = []

    def finish(self):
        self._flush()

    # -- replay ------------------------------------------------------------------------------------------------------
    def replay(self):
        """Issue the recorded step.  The caller has made the recorded compute stream current."""
        replay = self.lib.psld_tape_replay
        for seg in self.segments:
            if seg[0] == "c":
                rc = replay(seg[2], seg[3], C.byref(self._failed))
                if rc:
                    _lib.check(rc, f"psld_tape_replay (entry {self._failed.value} of a {seg[3]}-entry segment)")
            else:
                seg[1]()

    def __del__(self):
        try:
            for ev in self._events:
                self.lib.psld_tape_event_destroy(ev)
        except Exception:
            pass


class _LibProxy:
    """Stands in for the loaded CDLL while a tape records: launching entry points are called AND noted."""

    def __init__(self, real, tape: LaunchTape):
        self._real, self._tape = real, tape

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in _lib.SIGNATURES and _lib.is_launch(name):
            idx = self._real.psld_tape_fn_index(name.encode())
            if idx < 0:
                raise RuntimeError(f"launch tape: libpsld_hip has no stub for {name} (stale tape_stubs.inc?)")
            tape = self._tape

            def recorded(*args, _fn=fn, _idx=idx, _name=name):
                if tape.through_stubs:      # self-test: the launch itself goes through psld_tape_replay's stub
                    tape.add_launch(_idx, _name, args)
                    fn_i, words = tape._cur[-1]
                    one = np.zeros(1, dtype=ENTRY)
                    one["fn"][0], one["nargs"][0] = fn_i, len(words)
                    one["a"][0, :len(words)] = np.array(words, dtype=np.uint64)
                    return tape.lib.psld_tape_replay(one.ctypes.data, 1, C.byref(tape._failed))
                rc = _fn(*args)
                if rc == 0:
                    tape.add_launch(_idx, _name, args)
                return rc
            fn = recorded
        setattr(self, name, fn)
        return fn


class _AtenRecorder(_pd.TorchDispatchMode):
    """Notes the kernels PyTorch itself launches inside the recorded region (autograd glue, scalar arithmetic on t)."""

    def __init__(self, tape: LaunchTape):
        super().__init__()
        self.tape = tape

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        name = func._schema.name
        if name in _NO_KERNEL:
            return out
        ins = [a for a in tree_flatten((args, kwargs))[0] if isinstance(a, torch.Tensor)]
        outs = [o for o in tree_flatten(out)[0] if isinstance(o, torch.Tensor)]
        if not any(t.is_cuda for t in ins + outs):
            return out
        tape = self.tape
        cur = torch.cuda.current_stream()
        if not tape.through_stubs and tape.stream is not None and cur != tape.stream:
            raise RuntimeError(f"launch tape: {name} ran on a stream other than the step's compute stream")
        label = str(func)
        if func._schema.is_mutable:
            tape.add_callable(lambda: func(*args, **kwargs), label, keep=ins)
            return out
        in_storages = {t.untyped_storage().data_ptr() for t in ins if t.is_cuda}
        if outs and all(o.untyped_storage().data_ptr() in in_storages for o in outs):
            return out                                          # a view this table does not name
        if len(outs) == 1 and outs[0] is out:
            dst = out
            tape.add_callable(lambda: dst.copy_(func(*args, **kwargs)), label, keep=ins + outs)
            tape.aten_outputs.append((label, outs, ins))
        else:
            def rerun():
                for d, r in zip(outs, [o for o in tree_flatten(func(*args, **kwargs))[0] if isinstance(o, torch.Tensor)]):
                    d.copy_(r)
            tape.add_callable(rerun, label, keep=ins + outs)
            tape.aten_outputs.append((label, outs, ins))
        return out


_active: List[LaunchTape] = []


def active() -> LaunchTape | None:
    return _active[-1] if _active else None


class record:
    """``with record(tape): ...`` - every libpsld_hip launch, stream edge and ATen kernel inside lands on ``tape``."""

    def __init__(self, tape: LaunchTape):
        self.tape = tape
        self._mode = _AtenRecorder(tape)

    def __enter__(self):
        if _active:
            raise RuntimeError("launch tape: recordings do not nest")
        self.tape.stream = torch.cuda.current_stream()
        _active.append(self.tape)
        _lib.set_proxy(_LibProxy(_lib.load_real(), self.tape))
        self._mode.__enter__()
        return self.tape

    def __exit__(self, *exc):
        self._mode.__exit__(*exc)
        _lib.set_proxy(None)
        _active.pop()
        self.tape.finish()
        return False


def note_edge(src: "torch.cuda.Stream", dst: "torch.cuda.Stream"):
    """Called where the executor orders ``dst`` after ``src`` with an event (score_fn.py flush_side / join_side)."""
    t = active()
    if t is not None:
        t.add_edge(src.cuda_stream, dst.cuda_stream)

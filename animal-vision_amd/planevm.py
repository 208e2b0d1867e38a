"""Plane programs: the host side of csrc/ew.hip and of the stage calls the UV species are made of.

The reference's UV species (animals/reindeer.py, goldfish.py, ...) are NumPy expression chains over HxW float32
planes with a blur / resize / Sobel / percentile here and there.  A species here is written ONCE against a small
"backend" vocabulary (`be.clip01`, `be.where`, `be.safe_norm`, `be.blur`, `be.percentile`, operators on values):

* `DeviceBackend` (this file, the product): values are nodes of an expression DAG; nothing runs while the species
  code executes.  Chains of elementwise operators are lowered to register programs and run as single fused launches
  (`avx_ew_run`); frame-wide reductions become accumulators whose results stay in a device-side scalar table; the
  non-elementwise stages are the C-ABI calls of include/avx.h.  The recorded call sequence is replayed per frame.
* `oracle/np_backend.py` (test infrastructure): the same vocabulary executed eagerly with NumPy and the oracle's
  OpenCV restatements, pinned against the reference's own outputs.

Lowering notes.  Python floats are weak (NEP 50): `0.35 * plane` is a float32 operation, as in the reference.
`x ** 2` is `x * x`, `x ** 0.5` is `sqrt` (NumPy's scalar-power fast paths); everything else is `powf`.
np.float64 scalars (which silently promote the reference's arrays to float64 under NumPy >= 2, e.g. goldfish.py:137)
are float32 constants here: ~1e-7 relative, far inside the 1e-4 contract."""
from __future__ import annotations

import ctypes
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import EW, EW_ACC, EW_PLANE, BandStackDesc, EwInsn, EwPlane, EwProgram, lib
from .dichromat import gaussian_taps
from .runtime import Context, DeviceBuffer, get_context
from .uv import bandpass_weights, fold_rgb_matrix, lobe_tables, uv_blur_ksize

_UNARY = {"neg": "NEG", "abs": "ABS", "sqrt": "SQRT", "exp": "EXP", "log": "LOG", "sin": "SIN", "cos": "COS", "floor": "FLOOR",
          "ceil": "CEIL", "clip01": "CLIP01", "tanh": "TANH", "not": "NOT"}
_BINARY = {"add": "ADD", "sub": "SUB", "mul": "MUL", "div": "DIV", "min": "MIN", "max": "MAX", "pow": "POW", "atan2": "ATAN2",
           "lt": "LT", "le": "LE", "gt": "GT", "ge": "GE", "eq": "EQ", "and": "AND", "or": "OR"}


class PlaneRef:
    """One addressable plane: element i of the frame at ptr[i * stride] (float32) / per-column / per-row vector."""

    __slots__ = ("buf", "offset", "stride", "kind", "uid")
    _next = 0

    def __init__(self, buf: DeviceBuffer, offset: int = 0, stride: int = 1, kind: str = "f32"):
        self.buf, self.offset, self.stride, self.kind = buf, int(offset), int(stride), kind
        PlaneRef._next += 1
        self.uid = PlaneRef._next

    @property
    def ptr(self) -> int:
        return self.buf.ptr + self.offset


class Val:
    """A node of the expression DAG (float32 per pixel; masks are 0.0 / 1.0)."""

    __slots__ = ("be", "op", "args", "imm", "_deps")
    __array_priority__ = 1000
    __array_ufunc__ = None  # NumPy scalars / arrays on the left defer to our reflected operators

    def __init__(self, be: "DeviceBackend", op: str, args: tuple = (), imm=None):
        self.be, self.op, self.args, self.imm, self._deps = be, op, args, imm, None

    # -- construction helpers ----------------------------------------------------------------------
    def _lift(self, other) -> "Val":
        if isinstance(other, Val):
            return other
        if isinstance(other, (bool, int, float, np.floating, np.integer)):
            return self.be.const(float(other))
        if isinstance(other, np.ndarray) and other.ndim == 0:
            return self.be.const(float(other))
        raise TypeError(f"plane programs cannot mix a plane with {type(other).__name__}")

    def _bin(self, op: str, other, swap: bool = False) -> "Val":
        o = self._lift(other)
        return Val(self.be, op, (o, self) if swap else (self, o))

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return Val(self.be, "neg", (self,))
    def __abs__(self): return Val(self.be, "abs", (self,))
    def __lt__(self, o): return self._bin("lt", o)
    def __le__(self, o): return self._bin("le", o)
    def __gt__(self, o): return self._bin("gt", o)
    def __ge__(self, o): return self._bin("ge", o)
    def __and__(self, o): return self._bin("and", o)
    def __or__(self, o): return self._bin("or", o)
    def __invert__(self): return Val(self.be, "not", (self,))

    def __pow__(self, p):
        if isinstance(p, Val):
            return self._bin("pow", p)
        p = float(p)
        if p == 2.0:
            return self * self
        if p == 1.0:
            return self
        if p == 0.5:
            return Val(self.be, "sqrt", (self,))
        if p == -1.0:
            return 1.0 / self
        return self._bin("pow", p)

    def __bool__(self):
        raise TypeError("a plane-program value has no truth value on the host: use be.where()")

    # leaves this value depends on: (plane uids, scalar slots)
    def deps(self) -> Tuple[frozenset, frozenset]:
        if self._deps is None:
            if self.op == "load":
                self._deps = (frozenset([self.imm.uid]), frozenset())
            elif self.op == "scalar":
                self._deps = (frozenset(), frozenset([self.imm]))
            else:
                p, s = frozenset(), frozenset()
                for a in self.args:
                    dp, ds = a.deps()
                    p, s = p | dp, s | ds
                self._deps = (p, s)
        return self._deps


def _f32_bits(v: float) -> int:
    return struct.unpack("<I", struct.pack("<f", float(np.float32(v))))[0]


class _Compiled:
    """One lowered elementwise program, kept alive for replay."""

    def __init__(self, be: "DeviceBackend", outputs: list):
        # outputs: ("store", Val, PlaneRef) | ("acc", Val, kind, slot)
        order: List[Val] = []
        seen: Dict[int, int] = {}
        leaves: Dict[tuple, Val] = {}

        def canon(v: Val) -> Val:  # one LOAD per plane, one SCALAR per slot; constants are re-materialised per use
            if v.op == "load":
                return leaves.setdefault(("p", v.imm.uid), v)
            if v.op == "scalar":
                return leaves.setdefault(("s", v.imm), v)
            return v

        # a constant that feeds a binary operator rides in that instruction (AVX_EW_IMM_A/B): never gets a register
        def imm_slot(node: Val) -> int:
            if node.op in _BINARY:
                a0, a1 = node.args
                if a1.op == "const" and a0.op != "const":
                    return 1
                if a0.op == "const" and a1.op != "const":
                    return 0
            return -1

        def visit(v: Val):  # iterative post-order; immediates are not visited (they get no register)
            v = canon(v)
            if id(v) in seen:
                return
            stack = [(v, 0)]
            while stack:
                node, k = stack.pop()
                node = canon(node)
                if id(node) in seen:
                    continue
                if k < len(node.args):
                    stack.append((node, k + 1))
                    if k == imm_slot(node):
                        continue
                    child = canon(node.args[k])
                    if id(child) not in seen:
                        stack.append((child, 0))
                else:
                    seen[id(node)] = len(order)
                    order.append(node)

        for o in outputs:
            visit(o[1])
        # last use (position in `order`; outputs are consumed right after their node is computed)
        last: Dict[int, int] = {}
        for pos, node in enumerate(order):
            for k, a in enumerate(node.args):
                if k != imm_slot(node):
                    last[id(canon(a))] = pos
        out_of: Dict[int, list] = {}
        for o in outputs:
            out_of.setdefault(id(canon(o[1])), []).append(o)
        planes: List[PlaneRef] = []
        plane_idx: Dict[Tuple[int, str], int] = {}

        def pidx(ref: PlaneRef) -> int:
            key = (ref.uid, ref.kind)
            if key not in plane_idx:
                if len(planes) >= _lib.AVX_EW_MAX_PLANES:
                    raise ValueError("plane program touches too many planes: materialise an intermediate with be.mat()")
                plane_idx[key] = len(planes)
                planes.append(ref)
            return plane_idx[key]

        accs = [o for o in outputs if o[0] == "acc"]
        if len(accs) > _lib.AVX_EW_MAX_ACC:
            raise ValueError("too many reductions in one plane program")
        free = list(range(_lib.AVX_EW_MAX_REGS - len(accs) - 1, -1, -1))  # pop() hands out r0 first
        acc_reg = {id(o): 0x80 + k for k, o in enumerate(accs)}  # placeholders: placed just above the temporaries below
        reg: Dict[int, int] = {}
        insn: List[Tuple[int, int, int, int, int]] = []
        for pos, node in enumerate(order):
            islot = imm_slot(node)
            srcs = [0 if k == islot else reg[id(canon(a))] for k, a in enumerate(node.args)]
            # registers of operands whose last use is this instruction can be reused for the result
            # (the interpreter reads every source before it writes the destination)
            for a in {id(canon(a)): canon(a) for k, a in enumerate(node.args) if k != islot}.values():
                if last.get(id(a)) == pos:
                    free.append(reg[id(a)])
            if not free:
                raise ValueError("plane program needs more than %d live registers: materialise an intermediate with be.mat()" % _lib.AVX_EW_MAX_REGS)
            free.sort(reverse=True)  # lowest register first: small programs run on the 16-register kernel
            r = free.pop()
            reg[id(node)] = r
            if node.op == "const":
                insn.append((EW["CONST"], r, 0, 0, _f32_bits(node.imm)))
            elif node.op == "scalar":
                insn.append((EW["SCALAR"], r, 0, 0, int(node.imm)))
            elif node.op == "load":
                insn.append((EW["LOAD"], r, 0, 0, pidx(node.imm)))
            elif node.op in _UNARY:
                insn.append((EW[_UNARY[node.op]], r, srcs[0], 0, 0))
            elif node.op in _BINARY:
                if islot >= 0:
                    insn.append((EW[_BINARY[node.op]] | (0x40 if islot == 0 else 0x80), r, srcs[0], srcs[1], _f32_bits(node.args[islot].imm)))
                else:
                    insn.append((EW[_BINARY[node.op]], r, srcs[0], srcs[1], 0))
            elif node.op == "select":
                insn.append((EW["SELECT"], r, srcs[0], srcs[1], srcs[2]))
            else:
                raise ValueError(f"unknown node {node.op}")
            for o in out_of.get(id(node), ()):
                if o[0] == "store":
                    insn.append((EW["STORE"], 0, r, 0, pidx(o[2])))
                else:
                    op = {"min": "ACCMIN", "max": "ACCMAX", "sum": "ACCSUM", "mean": "ACCSUM"}[o[2]]
                    insn.append((EW[op], acc_reg[id(o)], r, 0, 0))
            if id(node) not in last:  # nobody reads it later (an output-only node): release at once
                free.append(r)
        top = max([t[1] for t in insn if t[1] < 0x80] + [t[2] for t in insn] + [t[3] for t in insn]
                  + [t[4] for t in insn if (t[0] & 0x3f) == EW["SELECT"]] + [0]) + 1
        acc_reg = {k: top + (v - 0x80) for k, v in acc_reg.items()}
        insn = [(op, top + (d - 0x80) if d >= 0x80 else d, a_, b_, imm) for (op, d, a_, b_, imm) in insn]
        self.n_regs = top + len(accs)
        if len(insn) > _lib.AVX_EW_MAX_INSN:
            raise ValueError(f"plane program has {len(insn)} instructions (> {_lib.AVX_EW_MAX_INSN}): materialise an intermediate with be.mat()")
        self.n_insn = len(insn)
        self.insn = (EwInsn * len(insn))(*[EwInsn(*t) for t in insn])
        self.planes = (EwPlane * max(1, len(planes)))(*[EwPlane(p.ptr, p.stride, EW_PLANE[p.kind]) for p in planes])
        self._keep = planes
        flat = []
        for o in accs:
            flat += [acc_reg[id(o)], EW_ACC[o[2]], int(o[3])]
        self.acc = (ctypes.c_int32 * max(1, len(flat)))(*flat)
        p = EwProgram()
        p.struct_size = ctypes.sizeof(EwProgram)
        p.H, p.W = be.H, be.W
        p.n_insn, p.insn_host = len(insn), ctypes.cast(self.insn, ctypes.POINTER(EwInsn))
        p.n_planes, p.planes_host = len(planes), ctypes.cast(self.planes, ctypes.POINTER(EwPlane))
        p.n_acc, p.acc_host = len(accs), ctypes.cast(self.acc, ctypes.POINTER(ctypes.c_int32))
        p.scalars_dev, p.n_scalars = be.scalars.ptr, be.N_SCALARS
        self.program = p


class DeviceBackend:
    """Records the device call sequence of one species for one frame size; `run()` replays it per frame."""

    N_SCALARS = 256
    name = "device"

    def __init__(self, H: int, W: int, ctx: Optional[Context] = None, float_frames: bool = False):
        """float_frames: the frames are float32 HxWx3 (sRGB in [0,1] or [0,255]); outputs are float32 too."""
        self.ctx = ctx or get_context()
        self.float_frames = bool(float_frames)
        self.H, self.W, self.n = int(H), int(W), int(H) * int(W)
        self.plan: List = []          # callables(stream)
        self.pending: List = []       # outputs of the elementwise program being assembled
        self._pend_planes: set = set()
        self._pend_slots: set = set()
        self.scalars = self.ctx.malloc(8 * self.N_SCALARS)
        self._slot = 0
        self._bufs: List[DeviceBuffer] = [self.scalars]
        self._keep: List = []         # ctypes structs / host arrays referenced by the recorded calls
        item = 4 if self.float_frames else 1
        self.d_in = self.ctx.malloc(3 * self.n * item)
        self.d_base = self.ctx.malloc(3 * self.n * item)
        self.d_out = self.ctx.malloc(3 * self.n * item)
        self._bufs += [self.d_in, self.d_base, self.d_out]
        self.n_programs = 0
        self.n_insn = 0
        self._memo: Dict = {}         # id(Val) / tuple of ids -> materialised loads (keeps the Vals alive)
        self.stages: List[str] = []   # human-readable trace of the recorded plan (tools/dbg_plan.py)

    # -- memory ------------------------------------------------------------------------------------
    def _alloc(self, nbytes: int) -> DeviceBuffer:
        b = self.ctx.malloc(nbytes)
        self._bufs.append(b)
        return b

    def new_planes(self, k: int = 1) -> List[PlaneRef]:
        b = self._alloc(4 * self.n * k)
        return [PlaneRef(b, 4 * self.n * i) for i in range(k)]

    def new_slot(self) -> int:
        if self._slot >= self.N_SCALARS:
            raise ValueError("scalar table exhausted")
        self._slot += 1
        return self._slot - 1

    def close(self):
        for b in self._bufs:
            b.free()
        self._bufs = []
        self.plan = []

    # -- leaves ------------------------------------------------------------------------------------
    def const(self, v: float) -> Val:
        return Val(self, "const", (), float(v))

    def load(self, ref: PlaneRef) -> Val:
        return Val(self, "load", (), ref)

    def scalar(self, slot: int) -> Val:
        return Val(self, "scalar", (), int(slot))

    def _vector(self, vec: np.ndarray, kind: str) -> Val:
        v = np.ascontiguousarray(vec, dtype=np.float32)
        assert v.shape == ((self.W,) if kind == "col" else (self.H,))
        b = self._alloc(v.nbytes)
        self.ctx.upload(v, b)
        return self.load(PlaneRef(b, 0, 1, kind))

    def col(self, vec: np.ndarray) -> Val:
        """A length-W vector broadcast down the rows (`vec[None, :]`)."""
        return self._vector(vec, "col")

    def row(self, vec: np.ndarray) -> Val:
        """A length-H vector broadcast along the rows (`vec[:, None]`)."""
        return self._vector(vec, "row")

    # -- elementwise vocabulary ------------------------------------------------------------------------
    def _v(self, x) -> Val:
        return x if isinstance(x, Val) else self.const(float(x))

    def _un(self, op, x): return Val(self, op, (self._v(x),))
    def sqrt(self, x): return self._un("sqrt", x)
    def exp(self, x): return self._un("exp", x)
    def log(self, x): return self._un("log", x)
    def sin(self, x): return self._un("sin", x)
    def cos(self, x): return self._un("cos", x)
    def floor(self, x): return self._un("floor", x)
    def tanh(self, x): return self._un("tanh", x)
    def abs(self, x): return self._un("abs", x)
    def clip01(self, x): return self._un("clip01", x)
    def minimum(self, a, b): return Val(self, "min", (self._v(a), self._v(b)))
    def maximum(self, a, b): return Val(self, "max", (self._v(a), self._v(b)))
    def arctan2(self, y, x): return Val(self, "atan2", (self._v(y), self._v(x)))
    def power(self, x, p): return Val(self, "pow", (self._v(x), self._v(p)))
    def where(self, m, a, b): return Val(self, "select", (self._v(m), self._v(a), self._v(b)))

    def clip(self, x, lo, hi):
        if isinstance(lo, (int, float)) and isinstance(hi, (int, float)) and float(lo) == 0.0 and float(hi) == 1.0:
            return self.clip01(x)
        return self.minimum(self.maximum(x, lo), hi)

    # -- program assembly ------------------------------------------------------------------------------
    def _depends_on_pending(self, v: Val) -> bool:
        p, s = v.deps()
        return bool(p & self._pend_planes) or bool(s & self._pend_slots)

    def flush(self):
        if not self.pending:
            return
        comp = _Compiled(self, self.pending)
        self._keep.append(comp)
        self.n_programs += 1
        self.n_insn += comp.n_insn
        ctx = self.ctx
        self.plan.append(lambda s, c=comp: ctx._check(lib.avx_ew_run(ctx._h, ctypes.byref(c.program), s)))
        self.stages.append(f"ew[{comp.n_insn} insn, {comp.n_regs} regs, {comp.program.n_planes} planes, {comp.program.n_acc} acc]")
        self.pending, self._pend_planes, self._pend_slots = [], set(), set()

    def _push(self, out):
        if self._depends_on_pending(out[1]):
            self.flush()
        self.pending.append(out)
        if out[0] == "store":
            self._pend_planes.add(out[2].uid)
        else:
            self._pend_slots.add(out[3])

    def store(self, v, ref: PlaneRef):
        self._push(("store", self._v(v), ref))

    def mat(self, v) -> Val:
        """Force a value into a plane in HBM (a leaf for later programs); a value is materialised at most once."""
        v = self._v(v)
        if v.op == "load" and v.imm.kind == "f32":
            return v
        hit = self._memo.get(id(v))
        if hit is not None:
            return hit[1]
        ref = self.new_planes(1)[0]
        self.store(v, ref)
        out = self.load(ref)
        self._memo[id(v)] = (v, out)  # holds v: ids stay unique
        return out

    def mat_all(self, vals: Sequence) -> List[Val]:
        """mat() of several values, into consecutive planes (a following blur takes them without a copy)."""
        return self.mat_many(vals)[0]

    def mat_many(self, vals: Sequence) -> Tuple[List[Val], DeviceBuffer]:
        """Materialise into CONSECUTIVE planes of one buffer (for multi-plane blurs / percentiles)."""
        vals = [self._v(v) for v in vals]
        key = tuple(id(v) for v in vals)
        hit = self._memo.get(key)
        if hit is not None:
            return hit[1], hit[2]
        refs = self.new_planes(len(vals))
        for v, r in zip(vals, refs):
            self.store(v, r)
        out = [self.load(r) for r in refs]
        self._memo[key] = (vals, out, refs[0].buf)
        return out, refs[0].buf

    def _reduce(self, v, kind: str) -> Val:
        slot = self.new_slot()
        self._push(("acc", self._v(v), kind, slot))
        return self.scalar(slot)

    def min(self, v): return self._reduce(v, "min")
    def max(self, v): return self._reduce(v, "max")
    def sum(self, v): return self._reduce(v, "sum")
    def mean(self, v): return self._reduce(v, "mean")

    def safe_norm(self, x):
        """uv_helpers.py:47-53."""
        mn, mx = self.min(x), self.max(x)
        rng = mx - mn
        return self.where(rng < 1e-9, 0.0, (x - mn) / rng)

    # -- non-elementwise stages ------------------------------------------------------------------------
    def _call(self, fn, name: str = ""):
        self.flush()
        self.plan.append(fn)
        import inspect

        self.stages.append(name or inspect.stack()[1].function)

    def percentile(self, v, q: float) -> Val:
        """np.percentile over one value or over a list of values taken together (e.g. an HxWx3 array)."""
        vals = list(v) if isinstance(v, (list, tuple)) else [v]
        ptr, _ = self._contiguous(vals)
        count = self.n * len(vals)
        slot = self.new_slot()
        ctx, out = self.ctx, self.scalars.ptr + 8 * slot
        self._call(lambda s: ctx._check(lib.avx_percentile_dev(ctx._h, ptr, count, float(q), out, s)))
        return self.scalar(slot)

    def percentiles(self, vals: Sequence, q: float) -> List[Val]:
        """np.percentile(v, q) for several independent planes: resolved together (one set of radix passes per four)."""
        ptrs = [self._contiguous([v])[0] for v in vals]
        slots = [self.new_slot() for _ in vals]
        k = len(vals)
        a_ptr = (ctypes.c_void_p * k)(*ptrs)
        a_n = (ctypes.c_size_t * k)(*([self.n] * k))
        a_q = (ctypes.c_double * k)(*([float(q)] * k))
        a_out = (ctypes.c_void_p * k)(*[self.scalars.ptr + 8 * s_ for s_ in slots])
        self._keep += [a_ptr, a_n, a_q, a_out]
        ctx = self.ctx
        self._call(lambda s: ctx._check(lib.avx_percentiles_dev(ctx._h, k, a_ptr, a_n, a_q, a_out, s)))
        return [self.scalar(s_) for s_ in slots]

    def _contiguous(self, vals: Sequence) -> Tuple[int, List[Val]]:
        """Device pointer of K consecutive float32 planes holding `vals` (materialising if they are not already so)."""
        vals = [self._v(v) for v in vals]
        refs = [v.imm for v in vals if v.op == "load"]
        if len(refs) == len(vals) and all(r.kind == "f32" and r.stride == 1 and r.buf is refs[0].buf and r.offset == refs[0].offset + 4 * self.n * i
                                           for i, r in enumerate(refs)):
            return refs[0].ptr, vals
        ms, _ = self.mat_many(vals)
        return ms[0].imm.ptr, ms

    def blur_taps(self, vals: Sequence, ksize: int, taps: np.ndarray) -> List[Val]:
        """cv2.GaussianBlur with an explicit kernel on K planes (shared Gaussian contract, BORDER_REFLECT_101)."""
        if ksize > _lib.AVX_MAX_KSIZE:
            raise ValueError(f"blur ksize {ksize} > {_lib.AVX_MAX_KSIZE}")
        src, _ = self._contiguous(vals)
        K = len(vals)
        out = self.new_planes(K)
        self._keep.append(taps)
        ctx, H, W, dst = self.ctx, self.H, self.W, out[0].ptr
        tp = taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self._call(lambda s: ctx._check(lib.avx_planes_gaussian_blur(ctx._h, src, dst, K, H, W, int(ksize), tp, 0, s)))
        return [self.load(r) for r in out]

    def streak(self, vals: Sequence, params) -> List[Val]:
        """apply_anisotropic_acuity_blur_with_streak(stack(vals), *params) (animal_utils.py:147-172, quirk Q3) on the three
        planes of a linear-light frame; params = (y_center, sigma_streak, sigma_far, falloff)."""
        from .dichromat import STREAK_STRIDE, streak_row_tables

        assert len(vals) == 3
        src, _ = self._contiguous(vals)
        out = self.new_planes(3)
        rows = np.ascontiguousarray(streak_row_tables(self.H, *params), dtype=np.float32)
        self._keep.append(rows)
        ctx, H, W, dst = self.ctx, self.H, self.W, out[0].ptr
        rp = rows.ctypes.data_as(ctypes.c_void_p)
        self._call(lambda s: ctx._check(lib.avx_streak_planes_f32(ctx._h, src, dst, H, W, rp, STREAK_STRIDE, s)))
        return [self.load(r) for r in out]

    def blur(self, vals: Sequence, sigma: float) -> List[Val]:
        """uv_helpers.gaussian_blur (:66-73) per plane: identity for sigma <= 0, ksize = 2*ceil(3 sigma)+1."""
        if sigma <= 0:
            return list(vals)
        k = uv_blur_ksize(sigma)
        return self.blur_taps(vals, k, gaussian_taps(k, sigma))

    def sobel(self, v) -> Tuple[Val, Val]:
        """cv2.Sobel(ksize=3, REFLECT101) -> (gx, gy)."""
        src, _ = self._contiguous([v])
        gx, gy = self.new_planes(2)
        ctx, H, W = self.ctx, self.H, self.W
        self._call(lambda s: ctx._check(lib.avx_sobel3_plane(ctx._h, src, H, W, gx.ptr, gy.ptr, s)))
        return self.load(gx), self.load(gy)

    def remap(self, vals: Sequence, map_x, map_y, border_value: float = 0.0) -> List[Val]:
        """cv2.remap(stack(vals), map_x, map_y, INTER_LINEAR, BORDER_CONSTANT) with per-pixel float32 maps."""
        K = len(vals)
        src, _ = self._contiguous(vals)
        mx, _ = self._contiguous([map_x])
        my, _ = self._contiguous([map_y])
        out = self.new_planes(K)
        ctx, H, W, dst = self.ctx, self.H, self.W, out[0].ptr
        self._call(lambda s: ctx._check(lib.avx_remap_linear_planes(ctx._h, src, K, H, W, mx, my, dst, float(border_value), s)))
        return [self.load(r) for r in out]

    def down_up(self, vals: Sequence, h: int, w: int, interp_down: int, interp_up: int) -> List[Val]:
        """cv2.resize(cv2.resize(stack(vals), (w, h), interp_down), (W, H), interp_up): the mosaic / coarse-field idiom
        (morpho.py:84-92).  interp: 0 NEAREST, 1 LINEAR, 2 CUBIC, 3 AREA."""
        K = len(vals)
        hwc = self._alloc(4 * self.n * K)
        for c, v in enumerate(vals):
            self.store(self._v(v), PlaneRef(hwc, 4 * c, K))
        small = self._alloc(4 * h * w * K)
        out = self._alloc(4 * self.n * K)
        ctx, H, W = self.ctx, self.H, self.W
        self._call(lambda s: ctx._check(lib.avx_resize_hwc(ctx._h, hwc.ptr, 0, H, W, K, small.ptr, int(h), int(w), int(interp_down), s)))
        self._call(lambda s: ctx._check(lib.avx_resize_hwc(ctx._h, small.ptr, 0, int(h), int(w), K, out.ptr, H, W, int(interp_up), s)))
        return [self.load(PlaneRef(out, 4 * c, K)) for c in range(K)]

    # -- species skeleton ------------------------------------------------------------------------------
    def _linear_to_srgb(self, l):
        """uv_helpers.linear_to_srgb (:40-44) as an expression (float frames; uint8 frames use the threshold table)."""
        a = 0.055
        return self.where(l <= 0.0031308, l * 12.92, (1 + a) * self.power(self.maximum(l, 0.0), 1 / 2.4) - a)

    def front(self, image: np.ndarray, panorama_scale: float) -> List[Val]:
        """to_float01 + srgb_to_linear + panorama_warp; the baseline (input dtype) goes to d_base.  -> linear [R, G, B]."""
        H, W = self.H, self.W
        new_w = W
        if panorama_scale and panorama_scale != 1.0 and abs(panorama_scale - 1.0) >= 1e-3:
            new_w = max(2, int(round(W * panorama_scale)))
            if new_w < W:
                raise ValueError("panorama_scale < 1 is not a usable configuration of the reference (its centre crop changes the frame shape)")
        lin = self._alloc(4 * 3 * self.n)
        self.lin_hwc = lin
        ctx, d_in, d_base = self.ctx, self.d_in, self.d_base
        if not self.float_frames:
            self._call(lambda s: ctx._check(lib.avx_uv_front_u8(ctx._h, d_in.ptr, H, W, new_w, lin.ptr, d_base.ptr, s)))
            return [self.load(PlaneRef(lin, 4 * c, 3)) for c in range(3)]
        # float frames (uv_helpers.py:15-23,33-37): /255 and clip only when the frame's maximum exceeds 1.001
        y = [self.load(PlaneRef(d_in, 4 * c, 3)) for c in range(3)]
        mx = self.max(self.maximum(self.maximum(y[0], y[1]), y[2]))
        a = 0.055
        pre = self._alloc(4 * 3 * self.n) if new_w > W else lin
        for c in range(3):
            v = self.where(mx > 1.001, self.clip01(y[c] / 255.0), y[c])
            self.store(self.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4), PlaneRef(pre, 4 * c, 3))
        if new_w > W:
            self._call(lambda s: ctx._check(lib.avx_panorama_warp_f32(ctx._h, pre.ptr, H, W, new_w, lin.ptr, s)), "panorama")
        rgb = [self.load(PlaneRef(lin, 4 * c, 3)) for c in range(3)]
        for c in range(3):  # baseline = linear_to_srgb(clip(baseline_lin, 0, 1)).astype(dtype)
            self.store(self._linear_to_srgb(self.clip01(rgb[c])), PlaneRef(d_base, 4 * c, 3))
        return rgb

    def bands(self, lambdas: np.ndarray, bands: Sequence[Tuple[float, float]], hsi_scale: float) -> List[Val]:
        """classic_rgb_to_hsi(_scaled)(baseline_lin) then integrate_band per (lo, hi): raw band planes."""
        wts = np.ascontiguousarray(np.stack([bandpass_weights(lambdas, lo, hi) for lo, hi in bands]), dtype=np.float32)
        return self.spectral_planes(lambdas, wts, hsi_scale)

    def spectral_planes(self, lambdas: np.ndarray, wts: np.ndarray, hsi_scale: float, downsample: Optional[bool] = None) -> List[Val]:
        """K weighted sums over the wavelength axis of classic_rgb_to_hsi(_scaled)(self.lin_hwc): wts is K x B."""
        H, W = self.H, self.W
        K = wts.shape[0]
        wts = np.ascontiguousarray(wts, dtype=np.float32)
        gains, denom = lobe_tables(lambdas)
        M = fold_rgb_matrix(wts, lambdas)
        d = BandStackDesc()
        d.struct_size = ctypes.sizeof(BandStackDesc)
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
        d.n_bands, d.band_matrix_host, d.n_wavelengths, d.lobe_gains_host, d.lobe_denom, d.band_weights_host = K, fp(M), int(lambdas.size), fp(gains), denom, fp(wts)
        if (0.0 < hsi_scale < 1.0) if downsample is None else downsample:
            hs, ws = max(1, int(round(H * hsi_scale))), max(1, int(round(W * hsi_scale)))
            if (hs, ws) != (H, W):
                d.small_h, d.small_w = hs, ws
        self._keep += [wts, gains, M, d]
        stack = self._alloc(4 * self.n * K)
        ctx, lin = self.ctx, self.lin_hwc
        self._call(lambda s: ctx._check(lib.avx_band_stack(ctx._h, lin.ptr, H, W, ctypes.byref(d), stack.ptr, s)), "band_stack")
        self.last_stack = stack
        return [self.load(PlaneRef(stack, 4 * k, K)) for k in range(K)]

    def encode(self, rgb: Sequence):
        """from_float01(linear_to_srgb(clip(render, 0, 1)), dtype) -> d_out (HWC)."""
        for c, v in enumerate(rgb):
            if self.float_frames:
                self.store(self._linear_to_srgb(self.clip01(self._v(v))), PlaneRef(self.d_out, 4 * c, 3))
            else:
                self.store(self._v(v), PlaneRef(self.d_out, c, 3, "u8_enc"))
        self.flush()

    # -- replay ----------------------------------------------------------------------------------------
    def run(self, image: np.ndarray, stream=None) -> Tuple[np.ndarray, np.ndarray]:
        ctx = self.ctx
        dt = np.float32 if self.float_frames else np.uint8
        assert image.shape == (self.H, self.W, 3) and (image.dtype == np.uint8) == (not self.float_frames)
        ctx.upload(np.ascontiguousarray(image, dtype=dt), self.d_in, stream)  # float64 frames: x.astype(float32), the reference's first step
        s = ctx._s(stream)
        for fn in self.plan:
            fn(s)
        base = ctx.download(self.d_base, image.shape, dt, stream=stream, sync=False)
        out = ctx.download(self.d_out, image.shape, dt, stream=stream)
        if image.dtype != dt:  # `.astype(orig_dtype)` of from_float01
            base, out = base.astype(image.dtype), out.astype(image.dtype)
        return base, out

    def run_device(self, stream=None):
        """Replay on frames already in d_in (device-resident loop: bench, pipeline)."""
        s = self.ctx._s(stream)
        for fn in self.plan:
            fn(s)


class DeviceProbes:
    """Frame statistics a species branches on, computed on the device (8 bytes come back to the host)."""

    _cache: Dict[tuple, tuple] = {}

    @classmethod
    def median_luma(cls, image: np.ndarray) -> float:
        """float(np.median(0.2126 R + 0.7152 G + 0.0722 B)) of to_float01(image) (rat_uv.py:100-105)."""
        H, W = image.shape[:2]
        floats = image.dtype != np.uint8
        hit = cls._cache.get((H, W, floats))
        if hit is None:
            if len(cls._cache) >= 4:
                cls._cache.pop(next(iter(cls._cache)))[0].close()
            be = DeviceBackend(H, W, float_frames=floats)
            if floats:
                y = [be.load(PlaneRef(be.d_in, 4 * c, 3)) for c in range(3)]
                mx = be.max(be.maximum(be.maximum(y[0], y[1]), y[2]))
                ch = [be.where(mx > 1.001, be.clip01(v / 255.0), v) for v in y]
            else:
                ch = [be.load(PlaneRef(be.d_in, c, 3, "u8")) / 255.0 for c in range(3)]
            slot_val = be.percentile(0.2126 * ch[0] + 0.7152 * ch[1] + 0.0722 * ch[2], 50.0)
            be.flush()
            hit = (be, slot_val.imm)
            cls._cache[(H, W, floats)] = hit
        be, slot = hit
        ctx = be.ctx
        ctx.upload(np.ascontiguousarray(image, dtype=np.float32 if floats else np.uint8), be.d_in)
        be.run_device()
        return float(ctx.download(be.scalars.view(8 * slot, 8), (1,), np.float64)[0])


def run_planes(inputs: Sequence[np.ndarray], build, *, scalars: bool = False, ctx: Optional[Context] = None):
    """NumPy in / NumPy out around one recorded plan (the step-wise helper surface: animals/animal_utils.py, uv_mappers.py).

    inputs: HxW float arrays (converted to float32, the reference helpers' working precision); build(be, vals) -> a list
    of Vals, each returned as an HxW float32 array -- or, with scalars=True, a list whose Vals are frame-wide scalars
    (reductions / percentiles), returned as Python floats.  Everything O(pixels) runs on the device (csrc/ew.hip and the
    stage kernels); the plan is recorded per call (milliseconds: these are drop-in helpers, not the fused hot path)."""
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in inputs]
    H, W = arrs[0].shape
    assert all(a.shape == (H, W) for a in arrs), "planes must share one HxW shape"
    be = DeviceBackend(H, W, ctx=ctx, float_frames=True)
    try:
        refs = be.new_planes(len(arrs))
        for a, r in zip(arrs, refs):
            be.ctx.upload(a, r.buf.view(r.offset, a.nbytes))
        outs = build(be, [be.load(r) for r in refs])
        if scalars:
            slots = []
            for v in outs:
                v = be._v(v)
                if v.op != "scalar":
                    raise TypeError("scalars=True: build() must return reductions / percentiles")
                slots.append(v.imm)
            be.flush()
        else:
            mats = [be.mat(v + 0.0 if be._v(v).op in ("load", "scalar", "const") else v) for v in outs]
            be.flush()
        be.run_device()
        if scalars:
            return [float(be.ctx.download(be.scalars.view(8 * s, 8), (1,), np.float64)[0]) for s in slots]
        return [be.ctx.download(m.imm.buf.view(m.imm.offset, 4 * H * W), (H, W), np.float32) for m in mats]
    finally:
        be.close()

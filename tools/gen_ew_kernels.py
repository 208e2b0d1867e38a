#!/usr/bin/env python3
"""Generate csrc/ew_gen.hip: one straight-line kernel per recorded plane program.

The UV species' NumPy expression chains are lowered by animal_vision_amd/planevm.py to short register programs that
csrc/ew.hip::k_ew interprets an instruction at a time.  The STRUCTURE of those programs (opcodes, registers, which plane
each LOAD / STORE touches and of what kind, the accumulators) is fixed by the species' code; only immediates, scalar
slots and plane pointers change from call to call.  `AVX_EW_DUMP=<file>` makes avx_ew_run append the structure of every
program it runs (one line each); csrc/ew_programs.txt is that dump over the GPU test suite + every species at 1080p,
de-duplicated.  This script turns each line into a kernel with the same arithmetic (one IEEE float32 operation per
instruction, same order, same reductions) in which the instruction stream is code: registers are compiler-allocated,
loads are scheduled together, nothing is dispatched at run time.  avx_ew_run finds the kernel by two 64-bit hashes of
the structure bytes and falls back to the interpreter for a program it has never seen.

    python tools/gen_ew_kernels.py            # csrc/ew_programs.txt -> csrc/ew_gen.hip
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "animal-vision_amd", "csrc", "ew_programs.txt")
DST = os.path.join(ROOT, "animal-vision_amd", "csrc", "ew_gen.hip")

# include/avx.h
(CONST, SCALAR, LOAD, STORE, ADD, SUB, MUL, DIV, MIN, MAX, POW, ATAN2, NEG, ABS, SQRT, EXP, LOG, SIN, COS, FLOOR, CEIL, CLIP01, TANH,
 LT, LE, GT, GE, EQ, AND, OR, NOT, SELECT, ACCMIN, ACCMAX, ACCSUM) = range(1, 36)
IMM_A, IMM_B = 0x40, 0x80
P_F32, P_U8, P_U8_LUT, P_COL, P_ROW, P_U8_ENC = range(6)
ACC_MIN, ACC_MAX, ACC_SUM, ACC_MEAN = range(4)

BINARY = {ADD: "x + y", SUB: "x - y", MUL: "x * y", DIV: "x / y", MIN: "fminf(x, y)", MAX: "fmaxf(x, y)", POW: "powf(x, y)", ATAN2: "atan2f(x, y)",
          LT: "x < y ? 1.f : 0.f", LE: "x <= y ? 1.f : 0.f", GT: "x > y ? 1.f : 0.f", GE: "x >= y ? 1.f : 0.f", EQ: "x == y ? 1.f : 0.f",
          AND: "(x != 0.f && y != 0.f) ? 1.f : 0.f", OR: "(x != 0.f || y != 0.f) ? 1.f : 0.f"}
UNARY = {NEG: "-x", ABS: "fabsf(x)", SQRT: "__fsqrt_rn(x)", EXP: "expf(x)", LOG: "logf(x)", SIN: "sinf(x)", COS: "cosf(x)", FLOOR: "floorf(x)",
         CEIL: "ceilf(x)", CLIP01: "x < 0.f ? 0.f : (x > 1.f ? 1.f : x)", TANH: "tanhf(x)", NOT: "x != 0.f ? 0.f : 1.f"}
M64 = (1 << 64) - 1


def parse(line):
    head, insn, planes, accs = [s.strip() for s in line.split(":")]
    n_insn, n_planes, n_acc = map(int, head.split())
    ins = [tuple(int(v) for v in t.split(",")) for t in insn.split()]
    kinds = [int(v) for v in planes.split()]
    acc = [tuple(int(v) for v in t.split(",")) for t in accs.split()]
    assert len(ins) == n_insn and len(kinds) == n_planes and len(acc) == n_acc, line[:80]
    return ins, kinds, acc


def structure_hash(ins, kinds, acc):
    h1, h2 = 0xcbf29ce484222325, 0x84222325cbf29ce4

    def mix(b):
        nonlocal h1, h2
        h1 = ((h1 ^ b) * 0x100000001b3) & M64
        h2 = ((h2 ^ ((b + 0x9e) & 0xffffffff)) * 0x100000001b3) & M64
        h2 ^= h2 >> 29

    for op, d, a, b, simm in ins:
        for v in (op, d, a, b, simm & 0xff):
            mix(v)
    mix(0xff)
    for k in kinds:
        mix(k)
    mix(0xfe)
    for r, k in acc:
        mix(r)
        mix(k)
    return h1, h2


def emit(idx, ins, kinds, acc):
    acc_of = {r: k for k, (r, _) in enumerate(acc)}
    uses_xy = any((op & 0x3f) in (LOAD, STORE) and kinds[simm] in (P_COL, P_ROW) for op, _, _, _, simm in ins)
    uses_enc = any((op & 0x3f) == STORE and kinds[simm] == P_U8_ENC for op, _, _, _, simm in ins)
    L = []
    w = L.append
    w(f"// program {idx}: {len(ins)} instructions, {len(kinds)} planes, {len(acc)} reductions")
    w("template <int PX>")
    w(f"__global__ __launch_bounds__(kET) void k_ews_{idx}(const EwArgs a) {{")
    w("    const int tid = threadIdx.x;")
    w("    (void)tid;")
    if uses_enc:
        w("    __shared__ float thr[256];")
        w("    __shared__ uint8_t coarse[kCoarseTableBytes];")
        w("    for (int i = tid; i < 256; i += kET) thr[i] = a.thr[i];")
        w("    for (int i = tid; i < kCoarseTableBytes; i += kET) coarse[i] = a.coarse[i];")
        w("    __syncthreads();")
    if acc:
        w(f"    float acc[{len(acc)}][PX];")
        for k, (_, kind) in enumerate(acc):
            init = "INFINITY" if kind == ACC_MIN else ("-INFINITY" if kind == ACC_MAX else "0.f")
            w(f"    EWG_FOR acc[{k}][k] = {init};")
    # wave-uniform operands: immediates, scalar-table entries, plane descriptors (read once, scalar registers)
    for pc, (op, d, ra, rb, simm) in enumerate(ins):
        o = op & 0x3f
        if o == CONST or (op & (IMM_A | IMM_B)):
            w(f"    const float i{pc} = __uint_as_float(a.insn[{pc}].imm);")
        elif o == SCALAR:
            w(f"    const float i{pc} = (float)a.scalars[a.insn[{pc}].imm];")
    for j in range(len(kinds)):
        w(f"    const avx_ew_plane p{j} = a.planes[{j}];")
    w("    const uint32_t n = (uint32_t)a.n, W = (uint32_t)a.W;")
    w("    (void)W;")
    w("    for (uint32_t base = blockIdx.x * (kET * PX); base < n; base += gridDim.x * (kET * PX)) {")
    w("        uint32_t ii[PX];")
    w("        bool valid[PX];")
    w("        EWG_FOR { const uint32_t i = base + tid + k * kET; valid[k] = i < n; ii[k] = valid[k] ? i : n - 1; }")
    if uses_xy:
        w("        uint32_t px[PX], py[PX];")
        w("        EWG_FOR { py[k] = ii[k] / W; px[k] = ii[k] - py[k] * W; }")
    reg = {}

    def src(r):
        if r in acc_of and r not in reg:
            return f"acc[{acc_of[r]}]"
        return reg[r]

    for pc, (op, d, ra, rb, simm) in enumerate(ins):
        o = op & 0x3f
        ia, ib = bool(op & IMM_A), bool(op & IMM_B)
        v = f"v{pc}"
        if o in (CONST, SCALAR):
            w(f"        float {v}[PX]; EWG_FOR {v}[k] = i{pc};")
        elif o == LOAD:
            kind, p = kinds[simm], f"p{simm}"
            ex = {P_F32: f"((const float*){p}.ptr)[(size_t)ii[k] * {p}.stride]",
                  P_U8: f"(float)((const uint8_t*){p}.ptr)[(size_t)ii[k] * {p}.stride]",
                  P_U8_LUT: f"a.lut[((const uint8_t*){p}.ptr)[(size_t)ii[k] * {p}.stride]]",
                  P_COL: f"((const float*){p}.ptr)[px[k]]",
                  P_ROW: f"((const float*){p}.ptr)[py[k]]"}[kind]
            w(f"        float {v}[PX]; EWG_FOR {v}[k] = {ex};")
        elif o == STORE:
            kind, p, s = kinds[simm], f"p{simm}", src(ra)
            if kind == P_F32:
                w(f"        EWG_FOR if (valid[k]) ((float*){p}.ptr)[(size_t)ii[k] * {p}.stride] = {s}[k];")
            else:
                assert kind == P_U8_ENC
                w(f"        EWG_FOR if (valid[k]) ((uint8_t*){p}.ptr)[(size_t)ii[k] * {p}.stride] = (uint8_t)quantize_coarse<float, kCoarseNFix>({s}[k], thr, coarse, a.lo_key);")
            continue
        elif o in BINARY:
            x = f"i{pc}" if ia else f"{src(ra)}[k]"
            y = f"i{pc}" if ib else f"{src(rb)}[k]"
            w(f"        float {v}[PX]; EWG_FOR {{ const float x = {x}, y = {y}; {v}[k] = {BINARY[o]}; }}")
        elif o in UNARY:
            w(f"        float {v}[PX]; EWG_FOR {{ const float x = {src(ra)}[k]; {v}[k] = {UNARY[o]}; }}")
        elif o == SELECT:
            w(f"        float {v}[PX]; EWG_FOR {v}[k] = {src(ra)}[k] != 0.f ? {src(rb)}[k] : {src(simm)}[k];")
        elif o in (ACCMIN, ACCMAX, ACCSUM):
            k = acc_of[d]
            fn = {ACCMIN: "fminf(acc[%d][k], x)", ACCMAX: "fmaxf(acc[%d][k], x)", ACCSUM: "acc[%d][k] + x"}[o] % k
            w(f"        EWG_FOR {{ const float x = {src(ra)}[k]; acc[{k}][k] = valid[k] ? {fn} : acc[{k}][k]; }}")
            continue
        else:
            raise ValueError(f"opcode {o}")
        reg[d] = v
    w("    }")
    if acc:
        w(f"    ew_reduce_tail<{len(acc)}, PX>(a, acc);")
    w("}")
    return "\n".join(L)


def main():
    lines = sorted({l.strip() for l in open(SRC) if l.strip()})
    progs = []
    for l in lines:
        ins, kinds, acc = parse(l)
        progs.append((structure_hash(ins, kinds, acc), ins, kinds, acc))
    progs.sort(key=lambda t: t[0])
    keys = [p[0] for p in progs]
    assert len(set(keys)) == len(keys), "hash collision between recorded programs"
    out = ["// GENERATED by tools/gen_ew_kernels.py from csrc/ew_programs.txt -- do not edit; regenerate instead.",
           "// One straight-line kernel per recorded plane program (see the generator's docstring); csrc/ew.hip::avx_ew_run dispatches by hash.",
           '#include "ew_common.h"', "", "using namespace avxk;", "using namespace avxew;", "",
           "#define EWG_FOR _Pragma(\"unroll\") for (int k = 0; k < PX; ++k)", "", "namespace {", ""]
    for i, (_, ins, kinds, acc) in enumerate(progs):
        out.append(emit(i, ins, kinds, acc))
        out.append("")
    out.append("}  // namespace")
    out.append("")
    out.append("namespace avxew {")
    out.append("const EwSpecEntry kEwSpec[] = {")
    for i, ((h1, h2), _, _, _) in enumerate(progs):
        out.append(f"    {{0x{h1:016x}ull, 0x{h2:016x}ull, k_ews_{i}<4>, k_ews_{i}<8>}},")
    out.append("};")
    out.append(f"const int kEwSpecCount = {len(progs)};")
    out.append("}  // namespace avxew")
    open(DST, "w").write("\n".join(out) + "\n")
    print(f"{len(progs)} programs -> {DST} ({sum(len(p[1]) for p in progs)} instructions)")


if __name__ == "__main__":
    main()

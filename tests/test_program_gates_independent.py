"""Independent check of every constraint PROGRAM: the programs in vectorx_amd/synth/synth_circuit.cpp are
run by both the oracle's interpreter and the GPU JIT, so byte-identical proofs only show that the two evaluators agree.
Here each program is interpreted in plain Python and compared, constraint by constraint, with the gate's DEFINING RELATION
written directly from plonky2's gate definitions (gates/{arithmetic_extension, base_sum, exponentiation, random_access,
multiplication_extension, reducing, reducing_extension, poseidon_mds, arithmetic_base}.rs `eval_unfiltered`) — on the
satisfying synthetic witness rows (all constraints zero) AND on random rows (non-zero values must match exactly, so a
program that merely vanishes on valid rows would be caught).  CosetInterpolationGate has its own Lagrange check in
tests/test_oracle_prover.py."""
import ctypes
import random

import pytest

from vectorx_amd.synth import SynthCircuit

P = 0xFFFFFFFF00000001
OP_END, OP_LDW, OP_LDC, OP_LDI, OP_ADD, OP_SUB, OP_MUL, OP_PUSH, OP_LDP = range(9)


def run_program(words, start, wires, consts, pih=(0, 0, 0, 0)):
    """interpret one constraint program (include/vxprover.h VX_OP_*) on a row: returns the pushed constraints"""
    R, out, pc = {}, [], start
    while True:
        ins = words[pc]
        op, dst, a, b = ins & 0xFF, (ins >> 8) & 0xFF, (ins >> 16) & 0xFFFF, (ins >> 32) & 0xFFFF
        if op == OP_END:
            return out
        if op == OP_LDW:
            R[dst] = wires[a]
        elif op == OP_LDC:
            R[dst] = consts[a]
        elif op == OP_LDI:
            pc += 1
            R[dst] = words[pc] % P
        elif op == OP_ADD:
            R[dst] = (R[a] + R[b]) % P
        elif op == OP_SUB:
            R[dst] = (R[a] - R[b]) % P
        elif op == OP_MUL:
            R[dst] = R[a] * R[b] % P
        elif op == OP_PUSH:
            out.append(R[a])
        elif op == OP_LDP:
            R[dst] = pih[a]
        else:
            raise AssertionError(f"opcode {op}")
        pc += 1


# ---- F_p^2 = F_p[X] / (X^2 - 7) ----
def emul(x, y):
    return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def eadd(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def esub(x, y):
    return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)


def escale(x, s):
    return (x[0] * s % P, x[1] * s % P)


def ext(w, i):
    return (w[i], w[i + 1])


# ---- the gates' defining relations (plonky2 v0.2.0 gates/*.rs eval_unfiltered, base-field expansion) ----
def arithmetic(w, c, num_ops=20):
    return [(w[4 * i + 3] - (w[4 * i] * w[4 * i + 1] * c[0] + w[4 * i + 2] * c[1])) % P for i in range(num_ops)]


def arithmetic_extension(w, c, num_ops=10):
    out = []
    for i in range(num_ops):
        m0, m1, ad, o = ext(w, 8 * i), ext(w, 8 * i + 2), ext(w, 8 * i + 4), ext(w, 8 * i + 6)
        out += list(esub(o, eadd(escale(emul(m0, m1), c[0]), escale(ad, c[1]))))
    return out


def base_sum(w, c, num_limbs=63, base=2):
    limbs = [w[1 + i] for i in range(num_limbs)]
    computed = sum(l * pow(base, i, P) for i, l in enumerate(limbs)) % P
    out = [(computed - w[0]) % P]
    for l in limbs:
        prod = 1
        for k in range(base):
            prod = prod * (l - k) % P
        out.append(prod)
    return out


def exponentiation(w, c, nbits=66):
    base, bits, output = w[0], [w[1 + i] for i in range(nbits)], w[1 + nbits]
    inter = [w[2 + nbits + i] for i in range(nbits)]
    out = []
    for i in range(nbits):
        prev = 1 if i == 0 else inter[i - 1] * inter[i - 1] % P
        cur = bits[nbits - 1 - i]
        out.append((prev * (cur * base + (1 - cur)) - inter[i]) % P)
    out.append((output - inter[nbits - 1]) % P)
    return out


def random_access(w, c, bits=4, copies=4, extra=2):
    vec, per = 1 << bits, (1 << bits) + 2
    routed = per * copies + extra
    out = []
    for cp in range(copies):
        idx, claimed = w[per * cp], w[per * cp + 1]
        items = [w[per * cp + 2 + i] for i in range(vec)]
        bs = [w[routed + cp * bits + i] for i in range(bits)]
        for b in bs:
            out.append(b * (b - 1) % P)
        out.append((sum(b << i for i, b in enumerate(bs)) - idx) % P)
        for b in bs:                                   # fold the list by the index bits, least significant first
            items = [(x + b * (y - x)) % P for x, y in zip(items[0::2], items[1::2])]
        out.append((items[0] - claimed) % P)
    for i in range(extra):
        out.append((c[i] - w[per * copies + i]) % P)
    return out


def mul_extension(w, c, num_ops=13):
    out = []
    for i in range(num_ops):
        out += list(esub(ext(w, 6 * i + 4), escale(emul(ext(w, 6 * i), ext(w, 6 * i + 2)), c[0])))
    return out


def reducing(w, c, num_coeffs, extension):
    cw = 2 if extension else 1
    start_accs = 6 + cw * num_coeffs
    output, alpha, acc = ext(w, 0), ext(w, 2), ext(w, 4)
    out = []
    for i in range(num_coeffs):
        coeff = ext(w, 6 + 2 * i) if extension else (w[6 + i], 0)
        computed = eadd(emul(acc, alpha), coeff)
        nxt = output if i == num_coeffs - 1 else ext(w, start_accs + 2 * i)
        out += list(esub(computed, nxt))
        acc = nxt
    return out


def poseidon_mds(w, c):
    C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    ins = [ext(w, 2 * i) for i in range(12)]
    out = []
    for r in range(12):
        acc = (0, 0)
        for i in range(12):
            acc = eadd(acc, escale(ins[(i + r) % 12], C[i]))
        if r == 0:
            acc = eadd(acc, escale(ins[0], 8))
        out += list(esub(ext(w, 24 + 2 * r), acc))
    return out


def _desc_arrays(sc):
    d = sc.desc
    words = list((ctypes.c_uint64 * d.programs_len).from_address(d.programs))
    offs = list((ctypes.c_int32 * d.num_gates).from_address(d.program_offsets))
    sels = list((ctypes.c_int32 * d.num_gates).from_address(d.selector_indices))
    return d, words, offs, sels


@pytest.mark.parametrize("flags", [15, 2])
def test_every_program_equals_its_gates_defining_relation(flags):
    sc = SynthCircuit(7, seed=77, poseidon_percent=40, flags=flags)
    d, words, offs, sels = _desc_arrays(sc)
    n = 1 << 7
    w = sc.witness()
    cs = [list((ctypes.c_uint64 * n).from_address(d.constants_sigmas + 8 * n * k)) for k in range(d.num_constants)]
    rc = sc.row_counts()
    # row ranges of the program gates in the generator's layout (after the Poseidon / arithmetic body)
    refs = []
    if flags & 2:
        refs.append(("arithmetic", lambda ww, cc: arithmetic(ww, cc, 20)))
    if flags & 1:
        refs += [("arithext", arithmetic_extension), ("basesum", base_sum)]
    if flags & 4:
        refs += [("exp", exponentiation), ("randacc", random_access)]
    if flags & 8:
        refs += [("mulext", mul_extension), ("reducing", lambda ww, cc: reducing(ww, cc, 43, False)),
                 ("reducingext", lambda ww, cc: reducing(ww, cc, 32, True)), ("mds", poseidon_mds)]
    # gate index of a row = the value of its selector polynomial; program gates have program_offsets >= 0
    def gate_of_row(r):
        for g in range(d.num_gates):
            if cs[sels[g]][r] == g:
                return g
        raise AssertionError("row without a gate")
    rows_of_gate = {}
    for r in range(n):
        rows_of_gate.setdefault(gate_of_row(r), []).append(r)
    prog_gates = [g for g in range(d.num_gates) if offs[g] >= 0]
    # map each program gate to its reference by trying the satisfying rows: exactly one reference may vanish there
    rng = random.Random(5)
    matched = set()
    for g in prog_gates:
        rows = rows_of_gate[g]
        assert rows, g
        hits = []
        for name, ref in refs:
            wires = [int(w[k, rows[0]]) for k in range(135)]
            consts = [int(cs[d.num_selectors + k][rows[0]]) for k in range(d.num_constants - d.num_selectors)]
            got = run_program(words, offs[g], wires, consts)
            try:
                want = ref(wires, consts)
            except IndexError:
                continue
            if len(want) == len(got) and got == want and all(v == 0 for v in got):
                hits.append((name, ref))
        if not hits:
            # CosetInterpolationGate: checked against Lagrange interpolation in tests/test_oracle_prover.py
            # (2 constraints for the shifted point + 2 x 4 for the two intermediate (eval, prod) pairs + 2 for the result)
            assert flags & 8 and len(run_program(words, offs[g], [0] * 135, [0] * 8)) == 12 and "coset" not in matched
            matched.add("coset")
            continue
        # random (unsatisfying) rows: the program must equal the defining relation as a POLYNOMIAL MAP, not only on valid rows;
        # this also disambiguates gates whose constraints vanish on the same row (e.g. an all-zero row)
        ok_names = []
        for name, ref in hits:
            same = True
            for _ in range(6):
                wires = [rng.randrange(P) for _ in range(135)]
                consts = [rng.randrange(P) for _ in range(8)]
                got, want = run_program(words, offs[g], wires, consts), ref(wires, consts)
                same = same and got == want and any(v for v in got)
            if same:
                ok_names.append(name)
        assert len(ok_names) == 1, (g, [h[0] for h in hits], ok_names)
        matched.add(ok_names[0])
        for r in rows:                                   # and every synthetic row of the gate satisfies it
            wires = [int(w[k, r]) for k in range(135)]
            consts = [int(cs[d.num_selectors + k][r]) for k in range(d.num_constants - d.num_selectors)]
            assert all(v == 0 for v in run_program(words, offs[g], wires, consts)), (g, r)
    assert matched - {"coset"} == {name for name, _ in refs}, matched
    assert ("coset" in matched) == bool(flags & 8)

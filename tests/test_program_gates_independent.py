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


# ---- plonky2-u32 gates (gates/{arithmetic_u32, add_many_u32, subtraction_u32, range_check_u32, comparison}.rs) ----
def _limb4(l):
    return l * (l - 1) * (l - 2) * (l - 3) % P


def _horner(limbs, base):
    acc = 0
    for l in reversed(limbs):
        acc = (acc * base + l) % P
    return acc


def u32_arithmetic(w, c, num_ops=3):
    out = []
    for i in range(num_ops):
        m0, m1, ad, lo, hi, inverse = (w[6 * i + k] for k in range(6))
        out.append((inverse * (0xFFFFFFFF - hi) - 1) * lo % P)               # high == u32::MAX  =>  low == 0 (canonical 64-bit split)
        out.append((hi * (1 << 32) + lo - (m0 * m1 + ad)) % P)
        limbs = [w[6 * num_ops + 32 * i + j] for j in range(32)]
        out += [_limb4(l) for l in reversed(limbs)]
        out.append((_horner(limbs[:16], 4) - lo) % P)
        out.append((_horner(limbs[16:], 4) - hi) % P)
    return out


def u32_add_many(w, c, num_addends=3, num_ops=5):
    per, out = num_addends + 3, []
    for i in range(num_ops):
        total = sum(w[per * i + j] for j in range(num_addends + 1))           # addends + carry-in
        res, carry = w[per * i + num_addends + 1], w[per * i + num_addends + 2]
        out.append((carry * (1 << 32) + res - total) % P)
        limbs = [w[per * num_ops + 18 * i + j] for j in range(18)]
        out += [_limb4(l) for l in reversed(limbs)]
        out.append((_horner(limbs[:16], 4) - res) % P)
        out.append((_horner(limbs[16:], 4) - carry) % P)
    return out


def u32_subtraction(w, c, num_ops=6):
    out = []
    for i in range(num_ops):
        x, y, bin_, res, bout = (w[5 * i + k] for k in range(5))
        out.append((res - (x - y - bin_ + (1 << 32) * bout)) % P)
        limbs = [w[5 * num_ops + 16 * i + j] for j in range(16)]
        out += [_limb4(l) for l in reversed(limbs)]
        out.append((_horner(limbs, 4) - res) % P)
        out.append(bout * (1 - bout) % P)
    return out


def u32_range_check(w, c, n=7):
    out = []
    for i in range(n):
        aux = [w[n + 16 * i + j] for j in range(16)]
        out.append((_horner(aux, 4) - w[i]) % P)
        out += [_limb4(a) for a in aux]
    return out


def comparison(w, c, nc=16):
    first = [w[4 + k] for k in range(nc)]
    second = [w[4 + nc + k] for k in range(nc)]
    out = [(_horner(first, 4) - w[0]) % P, (_horner(second, 4) - w[1]) % P]
    so_far = 0
    for k in range(nc):
        out += [_limb4(first[k]), _limb4(second[k])]
        diff = (second[k] - first[k]) % P
        dummy, eq, inter = w[4 + 2 * nc + k], w[4 + 3 * nc + k], w[4 + 4 * nc + k]
        out.append((diff * dummy - (1 - eq)) % P)
        out.append(eq * diff % P)
        out.append((inter - eq * so_far) % P)
        so_far = (inter + (1 - eq) * diff) % P
    out.append((w[3] - so_far) % P)
    bits = [w[4 + 5 * nc + b] for b in range(3)]
    out += [b * (1 - b) % P for b in bits]
    out.append((4 + w[3] - _horner(bits, 2)) % P)
    out.append((w[2] - bits[2]) % P)
    return out


def _desc_arrays(sc):
    d = sc.desc
    words = list((ctypes.c_uint64 * d.programs_len).from_address(d.programs))
    offs = list((ctypes.c_int32 * d.num_gates).from_address(d.program_offsets))
    sels = list((ctypes.c_int32 * d.num_gates).from_address(d.selector_indices))
    return d, words, offs, sels


@pytest.mark.parametrize("flags", [15, 2, 32, 32 | 15])
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
    if flags & 32:
        refs += [("u32arith", u32_arithmetic), ("u32addmany", u32_add_many), ("u32sub", u32_subtraction), ("u32range", u32_range_check),
                 ("comparison", comparison)]
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


def test_u32_block_means_what_the_reference_circuit_means():
    """The witness rows of the U32 block are not just constraint-satisfying: they compute what verify_voting_threshold
    (/root/reference/circuits/builder/justification.rs:164-186) computes — num_signed = sum of the signed bits, the two scaled
    products, and is_valid = [num_active * 2 <= num_signed * 3] — checked here on plain Python integers."""
    sc = SynthCircuit(8, seed=5, poseidon_percent=40, flags=32)
    d, words, offs, sels = _desc_arrays(sc)
    n = 1 << 8
    w = sc.witness()
    cs = [list((ctypes.c_uint64 * n).from_address(d.constants_sigmas + 8 * n * k)) for k in range(d.num_constants)]
    u32 = sc.row_counts()["u32_each"]
    rows = {}
    for r in range(n):
        for g in range(d.num_gates):
            if cs[sels[g]][r] == g and offs[g] >= 0:
                rows.setdefault(g, []).append(r)
    by_len = {len(v): v for v in rows.values()}
    arith = by_len[u32 + 1]                       # the chained adds + the threshold row
    signed, total = 0, 0
    for r in arith[:-1]:
        for op in range(3):
            m0, m1, ad, lo, hi = (int(w[6 * op + k, r]) for k in range(5))
            assert (m0, m1, hi) == (signed, 1, 0) and ad in (0, 1) and lo == signed + ad
            signed, total = lo, total + 1
    t = arith[-1]
    assert [int(w[k, t]) for k in (0, 1, 3)] == [signed, 3, (3 * signed) & 0xFFFFFFFF]
    assert [int(w[6 + k, t]) for k in (0, 1, 3)] == [total, 2, (2 * total) & 0xFFFFFFFF]
    m0, m1, ad, lo, hi = (int(w[12 + k, t]) for k in range(5))
    assert hi * (1 << 32) + lo == m0 * m1 + ad and hi > 0
    cmp_rows = [v for v in rows.values() if len(v) == u32 and int(w[0, v[0]]) == 2 * total and int(w[1, v[0]]) == 3 * signed]
    assert len(cmp_rows) == 1 and int(w[2, cmp_rows[0][0]]) == int(2 * total <= 3 * signed) == 1
    for r in cmp_rows[0]:                          # every ComparisonGate row decides a <= b correctly
        assert int(w[2, r]) == int(int(w[0, r]) <= int(w[1, r]))

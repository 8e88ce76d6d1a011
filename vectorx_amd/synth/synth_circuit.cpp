// Synthetic circuit + witness generator — the CALLER side of the drop-in boundary.
//
// In the reference the prover is handed a compiled circuit and a finished witness by plonky2x
// (CircuitBuilder::build + witness generation: /root/reference/circuits/header_range.rs:144,167), both
// of which are out of scope for the GPU work.  The real header_range gate mix
// cannot be extracted without Rust, so benchmarks and parity tests use this DECLARED
// stand-in: a standard_recursion_config circuit (135 wires, 80 routed) over the gate set
//   NoopGate, ConstantGate{2}, PublicInputGate, ArithmeticGate{20 ops}, PoseidonGate
// (+ with VXS_FLAG_PROGRAM_GATES: ArithmeticExtensionGate{10 ops} and BaseSumGate<2>{63 limbs}, supplied to the
//  prover as CONSTRAINT PROGRAMS — the mechanism for every gate outside the native set) laid out as
//   row 0  PublicInputGate (wires 0..3 = public_inputs_hash)
//   row 1  ConstantGate    (constants 0 and 1)
//   row 2  PoseidonGate    hashing the 4 public inputs in-circuit (outputs 0..3 copy-constrained to row 0)
//   then   P PoseidonGate rows forming a hash chain (output of one row copy-constrained to the next input),
//          A ArithmeticGate rows (20 ops each, op k's first multiplicand copy-constrained to op k-1's output),
//          [E ArithmeticExtensionGate rows, B BaseSumGate rows,] and NoopGate padding to 2^degree_bits rows.
// The gate list is sorted by (degree, id) and selectors are grouped exactly as plonky2's
// gates/selectors.rs::selector_polynomials does (max_degree = quotient_degree_factor + 1 = 9).
// Everything is derived deterministically from (degree_bits, seed, poseidon_percent, witness_seed, flags):
// `seed` fixes the CIRCUIT (layout, arithmetic constants, copy constraints), `witness_seed` only the witness values.
// Built into vectorx_amd/libvxsynth.so (plain g++; no GPU code).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>
#include "../../include/vxprover.h"
#include "../csrc/host_field.h"

using namespace vxh;

#define VXS_FLAG_PROGRAM_GATES 1     /* add ArithmeticExtensionGate + BaseSumGate rows, evaluated through constraint programs */
#define VXS_FLAG_ARITH_AS_PROGRAM 2  /* hand the ArithmeticGate to the prover as a program instead of the native gate */
#define VXS_FLAG_MORE_PROGRAM_GATES 4 /* + ExponentiationGate{66 bits} (degree 4) and RandomAccessGate{bits 4} (degree 5): 3 selector groups */
#define VXS_FLAG_LOOKUP 16            /* + one lookup table: LookupTableGate rows holding it, LookupGate rows looking values up (gates/lookup*.rs) */
#define VXS_FLAG_RECURSION_GATES 8    /* + the rest of the recursive verifier's gate set as programs: MulExtensionGate, ReducingGate,
                                         ReducingExtensionGate, PoseidonMdsGate, CosetInterpolationGate{4 bits, degree 8} */

/* vxs_build5's mix_permille entries (rows of a gate per thousand rows of the trace) */
enum { VXS_MIX_POSEIDON, VXS_MIX_ARITHMETIC, VXS_MIX_ARITHMETIC_EXTENSION, VXS_MIX_BASE_SUM, VXS_MIX_EXPONENTIATION, VXS_MIX_RANDOM_ACCESS,
       VXS_MIX_MUL_EXTENSION, VXS_MIX_REDUCING, VXS_MIX_REDUCING_EXTENSION, VXS_MIX_POSEIDON_MDS, VXS_MIX_COSET_INTERPOLATION, VXS_MIX_LOOKUP,
       VXS_MIX_COUNT };
#define VXS_FLAG_U32_GATES 32         /* + plonky2-u32's gates as programs (what plonky2x's U32Variable add / mul / sub / gt and its range checks
                                         instantiate — /root/reference/circuits/builder/justification.rs:164-186, decoder.rs:39-92):
                                         U32ArithmeticGate{3 ops}, U32AddManyGate{3 addends, 5 ops}, U32SubtractionGate{6 ops},
                                         U32RangeCheckGate{7 limbs}, ComparisonGate{32 bits, 16 chunks}; laid out as a voting-threshold block */

namespace {
struct SplitMix {
  u64 s;
  u64 next() {
    u64 z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  u64 field() {
    for (;;) {
      u64 v = next();
      if (v < P) return v;
    }
  }
};

enum Key { K_NOOP, K_CONST, K_PI, K_ARITH, K_POSEIDON, K_ARITHEXT, K_BASESUM, K_EXP, K_RANDACC, K_MULEXT, K_REDUCING, K_REDUCINGEXT,
           K_POSEIDONMDS, K_COSETINTERP, K_LOOKUP, K_LOOKUPTABLE, K_U32ARITH, K_U32ADDMANY, K_U32SUB, K_U32RANGE, K_COMPARISON, K_COUNT };
struct GateInfo {
  int key, type, param, degree;
  std::string id;
};

struct Synth {
  int degree_bits;
  int quotient_degree_factor = 8;
  size_t n;
  std::vector<int32_t> gate_types, gate_params, selector_indices, group_starts, group_ends, program_offsets;
  std::vector<u64> programs;
  std::vector<u64> constants_sigmas;  // [num_constants + 80][n]
  std::vector<u64> k_is;
  std::vector<uint32_t> pi_rows, pi_cols;
  std::vector<u64> witness;  // [135][n]
  std::vector<u64> public_inputs;
  size_t n_poseidon = 0, n_arith = 0, n_noop = 0, n_arithext = 0, n_basesum = 0, n_exp = 0, n_randacc = 0, n_recursion = 0;
  size_t n_lookup = 0, n_lookup_table = 0, n_u32 = 0;
  size_t rows_by_key[K_COUNT] = {0};   // rows carrying each gate (counted where the selectors are written)
  std::vector<int32_t> lut_lens, lookup_rows;
  std::vector<uint16_t> lut_inputs, lut_outputs;
  std::vector<int32_t> key_of_gate;   // desc gate index -> Key
  vx_circuit_desc desc;
};

// Fill one PoseidonGate row (gates/poseidon.rs wire layout) for the given 12 inputs, swap = 0.
void fill_poseidon_row(u64* w /* witness base */, size_t n, size_t row, const u64* in, u64* out) {
  const int START_FULL_0 = 29, START_PARTIAL = 65, START_FULL_1 = 87;
  auto W = [&](int col) -> u64& { return w[(size_t)col * n + row]; };
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  auto mds = [&](u64* s) {
    u64 o[12];
    for (int k = 0; k < 12; ++k) {
      u128 acc = 0;
      for (int i = 0; i < 12; ++i) acc += (u128)C[i] * s[(i + k) % 12];
      if (k == 0) acc += (u128)8 * s[0];
      o[k] = reduce128(acc);
    }
    memcpy(s, o, sizeof o);
  };
  u64 s[12];
  for (int i = 0; i < 12; ++i) W(i) = s[i] = in[i];
  W(24) = 0;                                  // swap
  for (int i = 0; i < 4; ++i) W(25 + i) = 0;  // delta_i = swap * (rhs - lhs)
  int round = 0;
  for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * round + i]);
    if (r != 0)
      for (int i = 0; i < 12; ++i) W(START_FULL_0 + 12 * (r - 1) + i) = s[i];
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    mds(s);
    ++round;
  }
  for (int r = 0; r < 22; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * round + i]);
    W(START_PARTIAL + r) = s[0];
    s[0] = sbox(s[0]);
    mds(s);
    ++round;
  }
  for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * round + i]);
    for (int i = 0; i < 12; ++i) W(START_FULL_1 + 12 * r + i) = s[i];
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    mds(s);
    ++round;
  }
  for (int i = 0; i < 12; ++i) W(12 + i) = out[i] = s[i];
}

struct DSU {
  std::vector<uint32_t> p;
  explicit DSU(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t find(uint32_t x) {
    while (p[x] != x) {
      p[x] = p[p[x]];
      x = p[x];
    }
    return x;
  }
  void unite(uint32_t a, uint32_t b) {
    a = find(a), b = find(b);
    if (a != b) p[std::max(a, b)] = std::min(a, b);
  }
};

// ---- constraint programs (include/vxprover.h VX_OP_*) -----------------------------------------------------
struct Prog {
  std::vector<u64> w;
  void ins(int op, int dst, int a = 0, int b = 0) { w.push_back(VX_INS(op, dst, a, b)); }
  void ldi(int dst, u64 imm) {
    ins(VX_OP_LDI, dst);
    w.push_back(imm);
  }
};
// gates/arithmetic_base.rs: output - (m0*m1*c0 + addend*c1), num_ops times
Prog program_arithmetic(int num_ops) {
  Prog p;
  p.ins(VX_OP_LDC, 4, 0);
  p.ins(VX_OP_LDC, 5, 1);
  for (int i = 0; i < num_ops; ++i) {
    for (int k = 0; k < 4; ++k) p.ins(VX_OP_LDW, k, 4 * i + k);
    p.ins(VX_OP_MUL, 6, 0, 1);
    p.ins(VX_OP_MUL, 6, 6, 4);
    p.ins(VX_OP_MUL, 7, 2, 5);
    p.ins(VX_OP_ADD, 6, 6, 7);
    p.ins(VX_OP_SUB, 6, 3, 6);
    p.ins(VX_OP_PUSH, 0, 6);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/arithmetic_extension.rs (D = 2, expanded to base-field wires): out - ((m0*m1)*c0 + addend*c1) in F_p[X]/(X^2-7)
Prog program_arithmetic_extension(int num_ops) {
  Prog p;
  p.ins(VX_OP_LDC, 10, 0);
  p.ins(VX_OP_LDC, 11, 1);
  p.ldi(12, 7);
  for (int i = 0; i < num_ops; ++i) {
    for (int k = 0; k < 8; ++k) p.ins(VX_OP_LDW, k, 8 * i + k);  // m0 = (r0,r1) m1 = (r2,r3) addend = (r4,r5) out = (r6,r7)
    p.ins(VX_OP_MUL, 13, 0, 2);    // m0a*m1a
    p.ins(VX_OP_MUL, 14, 1, 3);    // m0b*m1b
    p.ins(VX_OP_MUL, 14, 14, 12);  // *7
    p.ins(VX_OP_ADD, 13, 13, 14);  // prod.a
    p.ins(VX_OP_MUL, 15, 0, 3);
    p.ins(VX_OP_MUL, 16, 1, 2);
    p.ins(VX_OP_ADD, 15, 15, 16);  // prod.b
    p.ins(VX_OP_MUL, 13, 13, 10);
    p.ins(VX_OP_MUL, 17, 4, 11);
    p.ins(VX_OP_ADD, 13, 13, 17);
    p.ins(VX_OP_SUB, 13, 6, 13);
    p.ins(VX_OP_PUSH, 0, 13);
    p.ins(VX_OP_MUL, 15, 15, 10);
    p.ins(VX_OP_MUL, 17, 5, 11);
    p.ins(VX_OP_ADD, 15, 15, 17);
    p.ins(VX_OP_SUB, 15, 7, 15);
    p.ins(VX_OP_PUSH, 0, 15);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/base_sum.rs (B = 2): reduce_with_powers(limbs, 2) - sum, then limb*(limb-1) per limb
Prog program_base_sum(int num_limbs) {
  Prog p;
  p.ldi(1, 2);
  p.ldi(2, 1);
  p.ins(VX_OP_LDW, 3, num_limbs);  // top limb = wire 1 + (num_limbs-1)
  for (int i = num_limbs - 2; i >= 0; --i) {
    p.ins(VX_OP_MUL, 3, 3, 1);
    p.ins(VX_OP_LDW, 4, 1 + i);
    p.ins(VX_OP_ADD, 3, 3, 4);
  }
  p.ins(VX_OP_LDW, 4, 0);
  p.ins(VX_OP_SUB, 3, 3, 4);
  p.ins(VX_OP_PUSH, 0, 3);
  for (int i = 0; i < num_limbs; ++i) {
    p.ins(VX_OP_LDW, 4, 1 + i);
    p.ins(VX_OP_SUB, 5, 4, 2);
    p.ins(VX_OP_MUL, 5, 5, 4);
    p.ins(VX_OP_PUSH, 0, 5);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/exponentiation.rs: wire 0 base, wires 1..=nb power bits (LE), wire nb+1 output, then nb intermediate values
Prog program_exponentiation(int nb) {
  Prog p;
  p.ldi(1, 1);                      // r1 = 1
  p.ins(VX_OP_LDW, 2, 0);           // r2 = base
  for (int i = 0; i < nb; ++i) {
    if (i == 0) p.ins(VX_OP_MUL, 3, 1, 1);  // r3 = 1 (prev_intermediate for i = 0)
    else {
      p.ins(VX_OP_LDW, 3, 2 + nb + (i - 1));
      p.ins(VX_OP_MUL, 3, 3, 3);    // intermediate[i-1]^2
    }
    p.ins(VX_OP_LDW, 4, 1 + (nb - i - 1));  // cur_bit (bits are LE, accumulated BE)
    p.ins(VX_OP_SUB, 5, 1, 4);      // not_cur_bit
    p.ins(VX_OP_MUL, 6, 4, 2);      // cur_bit * base
    p.ins(VX_OP_ADD, 6, 6, 5);
    p.ins(VX_OP_MUL, 6, 6, 3);      // computed intermediate
    p.ins(VX_OP_LDW, 7, 2 + nb + i);
    p.ins(VX_OP_SUB, 6, 6, 7);
    p.ins(VX_OP_PUSH, 0, 6);
  }
  p.ins(VX_OP_LDW, 3, 1 + nb);
  p.ins(VX_OP_LDW, 4, 2 + nb + nb - 1);
  p.ins(VX_OP_SUB, 3, 3, 4);
  p.ins(VX_OP_PUSH, 0, 3);
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/random_access.rs: per copy [access_index, claimed, 2^bits items]; extra constants; bit wires after the routed ones
Prog program_random_access(int bits, int copies, int extra) {
  Prog p;
  const int vec = 1 << bits, per = 2 + vec, routed = per * copies + extra;
  p.ldi(1, 1);
  for (int c = 0; c < copies; ++c) {
    for (int i = 0; i < bits; ++i) {             // b (b - 1)
      p.ins(VX_OP_LDW, 40 + i, routed + c * bits + i);
      p.ins(VX_OP_SUB, 2, 40 + i, 1);
      p.ins(VX_OP_MUL, 2, 2, 40 + i);
      p.ins(VX_OP_PUSH, 0, 2);
    }
    p.ins(VX_OP_ADD, 3, 40 + bits - 1, 40 + bits - 1);  // reconstructed index: fold from the top bit: acc = 2*acc + b
    p.ins(VX_OP_SUB, 3, 3, 40 + bits - 1);              // r3 = top bit
    for (int i = bits - 2; i >= 0; --i) {
      p.ins(VX_OP_ADD, 3, 3, 3);
      p.ins(VX_OP_ADD, 3, 3, 40 + i);
    }
    p.ins(VX_OP_LDW, 4, per * c);
    p.ins(VX_OP_SUB, 3, 3, 4);
    p.ins(VX_OP_PUSH, 0, 3);
    for (int i = 0; i < vec; ++i) p.ins(VX_OP_LDW, 8 + i, per * c + 2 + i);
    int len = vec;
    for (int b = 0; b < bits; ++b) {             // list'[k] = x + bit_b (y - x)
      for (int k = 0; k < len / 2; ++k) {
        p.ins(VX_OP_SUB, 5, 8 + 2 * k + 1, 8 + 2 * k);
        p.ins(VX_OP_MUL, 5, 5, 40 + b);
        p.ins(VX_OP_ADD, 8 + k, 8 + 2 * k, 5);
      }
      len /= 2;
    }
    p.ins(VX_OP_LDW, 4, per * c + 1);
    p.ins(VX_OP_SUB, 5, 8, 4);
    p.ins(VX_OP_PUSH, 0, 5);
  }
  for (int i = 0; i < extra; ++i) {
    p.ins(VX_OP_LDC, 2, i);
    p.ins(VX_OP_LDW, 3, per * copies + i);
    p.ins(VX_OP_SUB, 2, 2, 3);
    p.ins(VX_OP_PUSH, 0, 2);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// ---- extension-field helpers for the emitters: an F_p^2 value lives in two consecutive registers (a, a+1) -------
// dst = x * y in F_p[X]/(X^2 - 7); r7 holds the constant 7; t, t+1 are scratch; dst may alias neither x nor y.
void emit_ext_mul(Prog& p, int dst, int x, int y, int r7, int t) {
  p.ins(VX_OP_MUL, dst, x, y);          // x.a y.a
  p.ins(VX_OP_MUL, t, x + 1, y + 1);    // x.b y.b
  p.ins(VX_OP_MUL, t, t, r7);
  p.ins(VX_OP_ADD, dst, dst, t);
  p.ins(VX_OP_MUL, dst + 1, x, y + 1);
  p.ins(VX_OP_MUL, t, x + 1, y);
  p.ins(VX_OP_ADD, dst + 1, dst + 1, t);
}
void emit_ext_push_diff(Prog& p, int x, int y, int t) {  // constraints x - y (two base-field constraints)
  p.ins(VX_OP_SUB, t, x, y);
  p.ins(VX_OP_PUSH, 0, t);
  p.ins(VX_OP_SUB, t, x + 1, y + 1);
  p.ins(VX_OP_PUSH, 0, t);
}
// gates/multiplication_extension.rs: output - m0 * m1 * c0, num_ops times (wires 6i.. : m0, m1, output)
Prog program_mul_extension(int num_ops) {
  Prog p;
  p.ldi(60, 7);
  p.ins(VX_OP_LDC, 59, 0);
  for (int i = 0; i < num_ops; ++i) {
    for (int k = 0; k < 6; ++k) p.ins(VX_OP_LDW, k, 6 * i + k);  // m0 = (r0,r1) m1 = (r2,r3) out = (r4,r5)
    emit_ext_mul(p, 6, 0, 2, 60, 10);
    p.ins(VX_OP_MUL, 6, 6, 59);
    p.ins(VX_OP_MUL, 7, 7, 59);
    emit_ext_push_diff(p, 4, 6, 10);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/reducing.rs (base-field coefficients) / reducing_extension.rs (F_p^2 coefficients):
//   wires: output 0..2, alpha 2..4, old_acc 4..6, coeffs from 6, then the accumulators (the last one IS the output);
//   constraints acc_prev * alpha + coeff_i - acc_i.
Prog program_reducing(int num_coeffs, bool ext_coeffs) {
  Prog p;
  p.ldi(60, 7);
  const int cw = ext_coeffs ? 2 : 1, start_accs = 6 + cw * num_coeffs;
  p.ins(VX_OP_LDW, 2, 2);
  p.ins(VX_OP_LDW, 3, 3);   // alpha = (r2, r3)
  p.ins(VX_OP_LDW, 4, 4);
  p.ins(VX_OP_LDW, 5, 5);   // acc = (r4, r5) = old_acc
  for (int i = 0; i < num_coeffs; ++i) {
    emit_ext_mul(p, 6, 4, 2, 60, 10);            // acc * alpha
    p.ins(VX_OP_LDW, 8, 6 + cw * i);
    p.ins(VX_OP_ADD, 6, 6, 8);
    if (ext_coeffs) {
      p.ins(VX_OP_LDW, 8, 6 + cw * i + 1);
      p.ins(VX_OP_ADD, 7, 7, 8);
    }
    const int aw = i == num_coeffs - 1 ? 0 : start_accs + 2 * i;
    p.ins(VX_OP_LDW, 4, aw);
    p.ins(VX_OP_LDW, 5, aw + 1);                 // next acc (a wire: keeps the degree at 2)
    emit_ext_push_diff(p, 6, 4, 10);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/poseidon_mds.rs: the MDS layer on 12 F_p^2 inputs (wires 0..24) -> outputs (wires 24..48), component-wise
Prog program_poseidon_mds() {
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  Prog p;
  for (int i = 0; i < 24; ++i) p.ins(VX_OP_LDW, i, i);
  for (int i = 0; i < 12; ++i) p.ldi(24 + i, C[i]);
  p.ldi(36, 8);
  for (int r = 0; r < 12; ++r)
    for (int comp = 0; comp < 2; ++comp) {
      p.ins(VX_OP_MUL, 40, 24, 2 * (r % 12) + comp);
      for (int i = 1; i < 12; ++i) {
        p.ins(VX_OP_MUL, 41, 24 + i, 2 * ((i + r) % 12) + comp);
        p.ins(VX_OP_ADD, 40, 40, 41);
      }
      if (r == 0) {
        p.ins(VX_OP_MUL, 41, 36, comp);
        p.ins(VX_OP_ADD, 40, 40, 41);
      }
      p.ins(VX_OP_LDW, 41, 24 + 2 * r + comp);
      p.ins(VX_OP_SUB, 40, 41, 40);
      p.ins(VX_OP_PUSH, 0, 40);
    }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/coset_interpolation.rs (subgroup_bits 4, degree 8): barycentric interpolation of 16 F_p^2 values on the coset
// shift*H_16, evaluated at a point, in chunks so that no constraint exceeds the degree:
//   wires: shift 0 | values 1..33 | evaluation_point 33,34 | evaluation_value 35,36 | intermediate_eval(i) 37+2i |
//          intermediate_prod(i) 37+2*NI+2i | shifted_evaluation_point 37+4*NI
//   shifted_point * shift = evaluation_point;  (eval, prod) <- (eval (z - x_j) + prod w_j v_j, prod (z - x_j)) over the
//   points of a chunk, starting from (0, 1) / the previous intermediates; the last eval is evaluation_value.
struct CosetInterp {
  static constexpr int BITS = 4, NP = 16, DEG = 8, NI = 2;
  u64 domain[NP], weights[NP];
  int chunk_end[NI + 1];
  CosetInterp() {
    const u64 g = root_of_unity(BITS);
    u64 x = 1;
    for (int j = 0; j < NP; ++j) domain[j] = x, x = mul(x, g);
    for (int j = 0; j < NP; ++j) {
      u64 d = 1;
      for (int k = 0; k < NP; ++k)
        if (k != j) d = mul(d, sub(domain[j], domain[k]));
      weights[j] = inv(d);
    }
    chunk_end[0] = DEG;                     // 8 points, then 7, then the last one
    chunk_end[1] = DEG + (DEG - 1);
    chunk_end[2] = NP;
  }
  static int w_shift() { return 0; }
  static int w_value(int j) { return 1 + 2 * j; }
  static int w_point() { return 33; }
  static int w_eval() { return 35; }
  static int w_ieval(int i) { return 37 + 2 * i; }
  static int w_iprod(int i) { return 37 + 2 * NI + 2 * i; }
  static int w_shifted() { return 37 + 4 * NI; }
};
Prog program_coset_interpolation() {
  CosetInterp G;
  Prog p;
  p.ldi(60, 7);
  p.ldi(42, 1);
  p.ins(VX_OP_LDW, 0, G.w_shift());
  p.ins(VX_OP_LDW, 2, G.w_shifted());
  p.ins(VX_OP_LDW, 3, G.w_shifted() + 1);        // z = (r2, r3)
  p.ins(VX_OP_LDW, 4, G.w_point());
  p.ins(VX_OP_LDW, 5, G.w_point() + 1);
  p.ins(VX_OP_MUL, 6, 2, 0);
  p.ins(VX_OP_MUL, 7, 3, 0);
  emit_ext_push_diff(p, 6, 4, 10);               // z * shift - evaluation_point
  p.ldi(20, 0);
  p.ldi(21, 0);                                   // eval = (r20, r21) = 0
  p.ldi(22, 1);
  p.ldi(23, 0);                                   // prod = (r22, r23) = 1
  int j = 0;
  for (int c = 0; c <= G.NI; ++c) {
    for (; j < G.chunk_end[c]; ++j) {
      p.ldi(30, G.domain[j]);
      p.ldi(31, G.weights[j]);
      p.ins(VX_OP_SUB, 32, 2, 30);
      p.ins(VX_OP_MUL, 33, 3, 42);                // (r32, r33) = z - x_j   (x_j is in the base field; r42 = 1: a register copy)
      p.ins(VX_OP_LDW, 34, G.w_value(j));
      p.ins(VX_OP_LDW, 35, G.w_value(j) + 1);
      p.ins(VX_OP_MUL, 34, 34, 31);
      p.ins(VX_OP_MUL, 35, 35, 31);               // w_j v_j
      emit_ext_mul(p, 36, 22, 34, 60, 10);        // prod * w_j v_j
      emit_ext_mul(p, 38, 20, 32, 60, 10);        // eval * (z - x_j)
      p.ins(VX_OP_ADD, 20, 38, 36);
      p.ins(VX_OP_ADD, 21, 39, 37);               // eval'
      emit_ext_mul(p, 40, 22, 32, 60, 10);        // prod'
      p.ins(VX_OP_MUL, 22, 40, 42);
      p.ins(VX_OP_MUL, 23, 41, 42);               // prod = prod'
    }
    if (c < G.NI) {
      p.ins(VX_OP_LDW, 44, G.w_ieval(c));
      p.ins(VX_OP_LDW, 45, G.w_ieval(c) + 1);
      p.ins(VX_OP_LDW, 46, G.w_iprod(c));
      p.ins(VX_OP_LDW, 47, G.w_iprod(c) + 1);
      emit_ext_push_diff(p, 44, 20, 10);          // intermediate_eval - eval
      emit_ext_push_diff(p, 46, 22, 10);          // intermediate_prod - prod
      p.ins(VX_OP_MUL, 20, 44, 42);
      p.ins(VX_OP_MUL, 21, 45, 42);
      p.ins(VX_OP_MUL, 22, 46, 42);
      p.ins(VX_OP_MUL, 23, 47, 42);               // continue from the wires: the degree starts again at 1
    } else {
      p.ins(VX_OP_LDW, 44, G.w_eval());
      p.ins(VX_OP_LDW, 45, G.w_eval() + 1);
      emit_ext_push_diff(p, 44, 20, 10);          // evaluation_value - eval
    }
  }
  p.ins(VX_OP_END, 0);
  return p;
}

// ---- plonky2-u32 gates (the crate behind plonky2x's U32Variable; un-vendored, restated from its gates/*.rs) --------------
// Shared pieces: a 2-bit limb range check  prod_{x<4} (limb - x)  (degree 4) and Horner recombination in base 4.
// Register conventions of these emitters: r51, r52, r53 = 1, 2, 3; r60 = 2^32; r61 = 4; r59 = 2^32 - 1.
void emit_u32_consts(Prog& p) {
  p.ldi(51, 1);
  p.ldi(52, 2);
  p.ldi(53, 3);
  p.ldi(59, 0xFFFFFFFFULL);
  p.ldi(60, (u64)1 << 32);
  p.ldi(61, 4);
}
// pushes limb (limb - 1)(limb - 2)(limb - 3) for the value in register `limb`; t, t+1 scratch
void emit_limb4_check(Prog& p, int limb, int t) {
  p.ins(VX_OP_SUB, t, limb, 51);
  p.ins(VX_OP_MUL, t, t, limb);
  p.ins(VX_OP_SUB, t + 1, limb, 52);
  p.ins(VX_OP_MUL, t, t, t + 1);
  p.ins(VX_OP_SUB, t + 1, limb, 53);
  p.ins(VX_OP_MUL, t, t, t + 1);
  p.ins(VX_OP_PUSH, 0, t);
}
// gates/arithmetic_u32.rs  U32ArithmeticGate: per op [m0, m1, addend, out_low, out_high, inverse] routed, 32 two-bit limbs after
// the routed block; constraints: canonicity (inverse (2^32-1 - high) - 1) low, high 2^32 + low - (m0 m1 + addend), the limb
// range checks from the top limb down, low / high recombination.
Prog program_u32_arithmetic(int num_ops) {
  Prog p;
  emit_u32_consts(p);
  for (int i = 0; i < num_ops; ++i) {
    for (int k = 0; k < 6; ++k) p.ins(VX_OP_LDW, k, 6 * i + k);
    p.ins(VX_OP_MUL, 6, 0, 1);
    p.ins(VX_OP_ADD, 6, 6, 2);          // computed_output
    p.ins(VX_OP_SUB, 7, 59, 4);         // u32::MAX - high
    p.ins(VX_OP_MUL, 7, 5, 7);
    p.ins(VX_OP_SUB, 7, 7, 51);         // hi_not_max
    p.ins(VX_OP_MUL, 7, 7, 3);
    p.ins(VX_OP_PUSH, 0, 7);            // hi_not_max_or_lo_zero
    p.ins(VX_OP_MUL, 8, 4, 60);
    p.ins(VX_OP_ADD, 8, 8, 3);
    p.ins(VX_OP_SUB, 8, 8, 6);
    p.ins(VX_OP_PUSH, 0, 8);            // combined_output - computed_output
    p.ldi(9, 0);
    p.ldi(10, 0);
    for (int j = 31; j >= 0; --j) {
      p.ins(VX_OP_LDW, 11, 6 * num_ops + 32 * i + j);
      emit_limb4_check(p, 11, 12);
      const int acc = j < 16 ? 9 : 10;
      p.ins(VX_OP_MUL, acc, acc, 61);
      p.ins(VX_OP_ADD, acc, acc, 11);
    }
    p.ins(VX_OP_SUB, 9, 9, 3);
    p.ins(VX_OP_PUSH, 0, 9);
    p.ins(VX_OP_SUB, 10, 10, 4);
    p.ins(VX_OP_PUSH, 0, 10);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/add_many_u32.rs  U32AddManyGate: per op [addends.., carry, out_result, out_carry] routed; 16 result + 2 carry limbs
Prog program_u32_add_many(int num_addends, int num_ops) {
  Prog p;
  emit_u32_consts(p);
  const int per = num_addends + 3;
  for (int i = 0; i < num_ops; ++i) {
    p.ins(VX_OP_LDW, 0, per * i);
    for (int j = 1; j <= num_addends; ++j) {   // the addends, then the carry-in
      p.ins(VX_OP_LDW, 1, per * i + j);
      p.ins(VX_OP_ADD, 0, 0, 1);
    }
    p.ins(VX_OP_LDW, 2, per * i + num_addends + 1);   // output_result
    p.ins(VX_OP_LDW, 3, per * i + num_addends + 2);   // output_carry
    p.ins(VX_OP_MUL, 4, 3, 60);
    p.ins(VX_OP_ADD, 4, 4, 2);
    p.ins(VX_OP_SUB, 4, 4, 0);
    p.ins(VX_OP_PUSH, 0, 4);
    p.ldi(5, 0);
    p.ldi(6, 0);
    for (int j = 17; j >= 0; --j) {
      p.ins(VX_OP_LDW, 7, per * num_ops + 18 * i + j);
      emit_limb4_check(p, 7, 8);
      const int acc = j < 16 ? 5 : 6;
      p.ins(VX_OP_MUL, acc, acc, 61);
      p.ins(VX_OP_ADD, acc, acc, 7);
    }
    p.ins(VX_OP_SUB, 5, 5, 2);
    p.ins(VX_OP_PUSH, 0, 5);
    p.ins(VX_OP_SUB, 6, 6, 3);
    p.ins(VX_OP_PUSH, 0, 6);
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/subtraction_u32.rs  U32SubtractionGate: per op [x, y, borrow, out_result, out_borrow] routed; 16 limbs of the result
Prog program_u32_subtraction(int num_ops) {
  Prog p;
  emit_u32_consts(p);
  for (int i = 0; i < num_ops; ++i) {
    for (int k = 0; k < 5; ++k) p.ins(VX_OP_LDW, k, 5 * i + k);
    p.ins(VX_OP_SUB, 5, 0, 1);
    p.ins(VX_OP_SUB, 5, 5, 2);          // result_initial = x - y - borrow
    p.ins(VX_OP_MUL, 6, 4, 60);
    p.ins(VX_OP_ADD, 6, 5, 6);
    p.ins(VX_OP_SUB, 6, 3, 6);
    p.ins(VX_OP_PUSH, 0, 6);            // out_result - (result_initial + 2^32 out_borrow)
    p.ldi(7, 0);
    for (int j = 15; j >= 0; --j) {
      p.ins(VX_OP_LDW, 8, 5 * num_ops + 16 * i + j);
      emit_limb4_check(p, 8, 9);
      p.ins(VX_OP_MUL, 7, 7, 61);
      p.ins(VX_OP_ADD, 7, 7, 8);
    }
    p.ins(VX_OP_SUB, 7, 7, 3);
    p.ins(VX_OP_PUSH, 0, 7);
    p.ins(VX_OP_SUB, 9, 51, 4);
    p.ins(VX_OP_MUL, 9, 4, 9);
    p.ins(VX_OP_PUSH, 0, 9);            // out_borrow (1 - out_borrow)
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/range_check_u32.rs  U32RangeCheckGate: input limbs 0..n, 16 two-bit aux limbs each
Prog program_u32_range_check(int num_input_limbs) {
  Prog p;
  emit_u32_consts(p);
  for (int i = 0; i < num_input_limbs; ++i) {
    p.ins(VX_OP_LDW, 0, i);
    p.ldi(1, 0);
    for (int j = 15; j >= 0; --j) {     // reduce_with_powers(aux_limbs, 4)
      p.ins(VX_OP_LDW, 2, num_input_limbs + 16 * i + j);
      p.ins(VX_OP_MUL, 1, 1, 61);
      p.ins(VX_OP_ADD, 1, 1, 2);
    }
    p.ins(VX_OP_SUB, 1, 1, 0);
    p.ins(VX_OP_PUSH, 0, 1);
    for (int j = 0; j < 16; ++j) {
      p.ins(VX_OP_LDW, 2, num_input_limbs + 16 * i + j);
      emit_limb4_check(p, 2, 3);
    }
  }
  p.ins(VX_OP_END, 0);
  return p;
}
// gates/comparison.rs  ComparisonGate{num_bits, num_chunks} (chunk_bits = 2): result_bool = [first_input <= second_input]
//   wires: first 0 | second 1 | result_bool 2 | most_significant_diff 3 | first chunks 4.. | second chunks | equality dummies |
//          chunks_equal | intermediate values | the chunk_bits + 1 bits of 2^chunk_bits + most_significant_diff
struct ComparisonLayout {
  int nc;
  int first(int c) const { return 4 + c; }
  int second(int c) const { return 4 + nc + c; }
  int eq_dummy(int c) const { return 4 + 2 * nc + c; }
  int chunks_equal(int c) const { return 4 + 3 * nc + c; }
  int intermediate(int c) const { return 4 + 4 * nc + c; }
  int msd_bit(int b) const { return 4 + 5 * nc + b; }
};
Prog program_comparison(int num_chunks) {
  Prog p;
  emit_u32_consts(p);
  const ComparisonLayout L{num_chunks};
  for (int side = 0; side < 2; ++side) {   // the chunks recombine to the inputs
    p.ldi(1, 0);
    for (int c = num_chunks - 1; c >= 0; --c) {
      p.ins(VX_OP_LDW, 2, side ? L.second(c) : L.first(c));
      p.ins(VX_OP_MUL, 1, 1, 61);
      p.ins(VX_OP_ADD, 1, 1, 2);
    }
    p.ins(VX_OP_LDW, 0, side);
    p.ins(VX_OP_SUB, 1, 1, 0);
    p.ins(VX_OP_PUSH, 0, 1);
  }
  p.ldi(10, 0);                            // most_significant_diff_so_far
  for (int c = 0; c < num_chunks; ++c) {
    p.ins(VX_OP_LDW, 2, L.first(c));
    p.ins(VX_OP_LDW, 3, L.second(c));
    emit_limb4_check(p, 2, 4);
    emit_limb4_check(p, 3, 4);
    p.ins(VX_OP_SUB, 6, 3, 2);             // difference
    p.ins(VX_OP_LDW, 7, L.eq_dummy(c));
    p.ins(VX_OP_LDW, 8, L.chunks_equal(c));
    p.ins(VX_OP_SUB, 9, 51, 8);            // 1 - chunks_equal
    p.ins(VX_OP_MUL, 4, 6, 7);
    p.ins(VX_OP_SUB, 4, 4, 9);
    p.ins(VX_OP_PUSH, 0, 4);               // difference * dummy - (1 - chunks_equal)
    p.ins(VX_OP_MUL, 4, 8, 6);
    p.ins(VX_OP_PUSH, 0, 4);               // chunks_equal * difference
    p.ins(VX_OP_LDW, 11, L.intermediate(c));
    p.ins(VX_OP_MUL, 4, 8, 10);
    p.ins(VX_OP_SUB, 4, 11, 4);
    p.ins(VX_OP_PUSH, 0, 4);               // intermediate - chunks_equal * so_far
    p.ins(VX_OP_MUL, 4, 9, 6);
    p.ins(VX_OP_ADD, 10, 11, 4);           // so_far = intermediate + (1 - chunks_equal) * difference
  }
  p.ins(VX_OP_LDW, 12, 3);
  p.ins(VX_OP_SUB, 4, 12, 10);
  p.ins(VX_OP_PUSH, 0, 4);                 // most_significant_diff - so_far
  for (int b = 0; b < 3; ++b) {
    p.ins(VX_OP_LDW, 13 + b, L.msd_bit(b));
    p.ins(VX_OP_SUB, 4, 51, 13 + b);
    p.ins(VX_OP_MUL, 4, 13 + b, 4);
    p.ins(VX_OP_PUSH, 0, 4);               // bit (1 - bit)
  }
  p.ins(VX_OP_ADD, 4, 15, 15);             // bits_combined = ((b2 * 2) + b1) * 2 + b0
  p.ins(VX_OP_ADD, 4, 4, 14);
  p.ins(VX_OP_ADD, 4, 4, 4);
  p.ins(VX_OP_ADD, 4, 4, 13);
  p.ins(VX_OP_ADD, 5, 61, 12);             // 2^chunk_bits + most_significant_diff
  p.ins(VX_OP_SUB, 4, 5, 4);
  p.ins(VX_OP_PUSH, 0, 4);
  p.ins(VX_OP_LDW, 5, 2);
  p.ins(VX_OP_SUB, 4, 5, 15);
  p.ins(VX_OP_PUSH, 0, 4);                 // result_bool - top bit
  p.ins(VX_OP_END, 0);
  return p;
}
}  // namespace

extern "C" {

typedef struct vxs_circuit vxs_circuit;

vxs_circuit* vxs_build4(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed, int flags, int quotient_degree_factor);
vxs_circuit* vxs_build5(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed, int flags, int quotient_degree_factor,
                        const int32_t* mix_permille);
vxs_circuit* vxs_build3(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed, int flags) {
  return vxs_build4(degree_bits, seed, poseidon_percent, witness_seed, flags, 8);
}
vxs_circuit* vxs_build2(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed) {
  return vxs_build3(degree_bits, seed, poseidon_percent, witness_seed, 0);
}
vxs_circuit* vxs_build(int degree_bits, uint64_t seed, int poseidon_percent) {
  return vxs_build3(degree_bits, seed, poseidon_percent, seed, 0);
}

// quotient_degree_factor (CircuitConfig::max_quotient_degree_factor; 8 in standard_recursion_config) sets the selector
// grouping (max_degree = qdf + 1: a gate of degree d shares a selector with its group, so d + 1 <= qdf + 1 once there is
// more than one group) and the partial-product chunking.  A gate family that does not fit is refused, except the
// PoseidonGate (degree 7): below 7 the circuit is built without it (row 2, which hashes the public inputs in a real
// circuit, becomes a NoopGate row that merely carries the hash).
vxs_circuit* vxs_build4(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed, int flags, int qdf) {
  return vxs_build5(degree_bits, seed, poseidon_percent, witness_seed, flags, qdf, nullptr);
}
// mix_permille (VXS_MIX_COUNT entries, or NULL = the fixed fractions of rounds 1-5): a DECLARED row mix — rows of each gate as
// parts per thousand of the trace, in the order of VXS_MIX_*; what is left is NoopGate padding (a real circuit is padded to a power
// of two the same way).  Every family with a non-zero entry must be enabled in `flags`; poseidon_percent is ignored.
vxs_circuit* vxs_build5(int degree_bits, uint64_t seed, int poseidon_percent, uint64_t witness_seed, int flags, int qdf,
                        const int32_t* mix) {
  if (degree_bits < 3 || degree_bits > 24 || poseidon_percent < 0 || poseidon_percent > 100 || qdf < 3 || qdf > 8) return nullptr;
  const bool has_poseidon = qdf >= 7;
  if (!has_poseidon) poseidon_percent = 0;
  if (((flags & VXS_FLAG_RECURSION_GATES) && qdf < 8) || ((flags & VXS_FLAG_MORE_PROGRAM_GATES) && qdf < 5)) return nullptr;
  const bool with_prog = flags & VXS_FLAG_PROGRAM_GATES, arith_prog = flags & VXS_FLAG_ARITH_AS_PROGRAM;
  const bool more_prog = flags & VXS_FLAG_MORE_PROGRAM_GATES, rec_prog = flags & VXS_FLAG_RECURSION_GATES;
  const bool with_lookup = flags & VXS_FLAG_LOOKUP, u32_prog = flags & VXS_FLAG_U32_GATES;
  if (u32_prog && (qdf < 5 || degree_bits < 5)) return nullptr;
  if ((with_prog && degree_bits < 4) || (more_prog && degree_bits < 5) || (rec_prog && degree_bits < 5) || (with_lookup && degree_bits < 5)) return nullptr;
  Synth* S = new Synth();
  S->degree_bits = degree_bits;
  S->quotient_degree_factor = qdf;
  const size_t n = S->n = (size_t)1 << degree_bits;
  const int NW = 135, NR = 80;
  SplitMix rng{seed};
  SplitMix wrng{witness_seed ^ 0xA5A5A5A55A5A5A5AULL};

  // ---- gate list sorted by (degree, id) as CircuitBuilder::build does; ids are plonky2's Debug strings ----
  std::vector<GateInfo> gates = {
      {K_NOOP, VX_GATE_NOOP, 0, 0, "NoopGate"},
      {K_CONST, VX_GATE_CONSTANT, 2, 1, "ConstantGate { num_consts: 2 }"},
      {K_PI, VX_GATE_PUBLIC_INPUT, 0, 1, "PublicInputGate"},
      {K_ARITH, arith_prog ? VX_GATE_PROGRAM : VX_GATE_ARITHMETIC, arith_prog ? 3 : 20, 3, "ArithmeticGate { num_ops: 20 }"},
      {K_POSEIDON, VX_GATE_POSEIDON, 0, 7, "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"},
  };
  if (!has_poseidon) gates.pop_back();
  if (with_prog) {
    gates.push_back({K_ARITHEXT, VX_GATE_PROGRAM, 3, 3, "ArithmeticExtensionGate { num_ops: 10 }"});
    gates.push_back({K_BASESUM, VX_GATE_PROGRAM, 2, 2, "BaseSumGate { num_limbs: 63 } + Base: 2"});
  }
  if (more_prog) {
    gates.push_back({K_EXP, VX_GATE_PROGRAM, 4, 4, "ExponentiationGate { num_power_bits: 66, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>"});
    gates.push_back({K_RANDACC, VX_GATE_PROGRAM, 5, 5, "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>"});
  }
  if (rec_prog) {
    gates.push_back({K_MULEXT, VX_GATE_PROGRAM, 3, 3, "MulExtensionGate { num_ops: 13 }"});
    gates.push_back({K_REDUCING, VX_GATE_PROGRAM, 2, 2, "ReducingGate { num_coeffs: 43 }"});
    gates.push_back({K_REDUCINGEXT, VX_GATE_PROGRAM, 2, 2, "ReducingExtensionGate { num_coeffs: 32 }"});
    gates.push_back({K_POSEIDONMDS, VX_GATE_PROGRAM, 1, 1, "PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"});
    gates.push_back({K_COSETINTERP, VX_GATE_PROGRAM, 8, 8, "CosetInterpolationGate { subgroup_bits: 4, degree: 8, barycentric_weights: [..], _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>"});
  }
  if (u32_prog) {
    const char* ph = ", _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>";
    gates.push_back({K_U32ARITH, VX_GATE_PROGRAM, 4, 4, std::string("U32ArithmeticGate { num_ops: 3") + ph});
    gates.push_back({K_U32ADDMANY, VX_GATE_PROGRAM, 4, 4, std::string("U32AddManyGate { num_addends: 3, num_ops: 5") + ph});
    gates.push_back({K_U32SUB, VX_GATE_PROGRAM, 4, 4, std::string("U32SubtractionGate { num_ops: 6") + ph});
    gates.push_back({K_U32RANGE, VX_GATE_PROGRAM, 4, 4, std::string("U32RangeCheckGate { num_input_limbs: 7") + ph});
    gates.push_back({K_COMPARISON, VX_GATE_PROGRAM, 4, 4, std::string("ComparisonGate { num_bits: 32, num_chunks: 16") + ph});
  }
  if (with_lookup) {
    gates.push_back({K_LOOKUP, VX_GATE_LOOKUP, 40, 0, "LookupGate {num_slots: 40, lut_hash: [..]}"});
    gates.push_back({K_LOOKUPTABLE, VX_GATE_LOOKUP_TABLE, 26, 0, "LookupTableGate {num_slots: 26, lut_hash: [..], last_lut_row: ..}"});
  }
  std::sort(gates.begin(), gates.end(), [](const GateInfo& a, const GateInfo& b) {
    return a.degree != b.degree ? a.degree < b.degree : a.id < b.id;
  });
  const int ng = (int)gates.size();
  int idx_of[K_COUNT];
  for (int k = 0; k < K_COUNT; ++k) idx_of[k] = -1;
  for (int g = 0; g < ng; ++g) idx_of[gates[g].key] = g;
  for (int g = 0; g < ng; ++g) S->key_of_gate.push_back(gates[g].key);
  // gates/selectors.rs::selector_polynomials, max_degree = quotient_degree_factor + 1 (9 in the standard configuration)
  const int max_degree = qdf + 1;
  std::vector<std::pair<int, int>> groups;
  if (gates.back().degree + ng - 1 <= max_degree) {
    groups.push_back({0, ng});
  } else {
    int start = 0;
    while (start < ng) {
      int size = 0;
      while (start + size < ng && size + gates[start + size].degree < max_degree) ++size;
      if (size == 0) { delete S; return nullptr; }  // a gate whose degree does not fit max_degree (checked above; plonky2 would not build it either)
      groups.push_back({start, start + size});
      start += size;
    }
  }
  // with lookups the lookup selectors (TransSre, TransLdc, InitSre, LastLdc + one "ends" selector per table) sit between
  // the gate selectors and the gate constants (gates/selectors.rs::selectors_lookup, selector_ends_lookups)
  const int NLS = with_lookup ? 4 + 1 : 0;
  const int NSEL = (int)groups.size(), NCONST = NSEL + NLS + 2;
  std::vector<int> group_of(ng);
  for (int g = 0; g < ng; ++g)
    for (int q = 0; q < NSEL; ++q)
      if (g >= groups[q].first && g < groups[q].second) group_of[g] = q;
  for (int g = 0; g < ng; ++g) {
    S->gate_types.push_back(gates[g].type);
    S->gate_params.push_back(gates[g].param);
    S->selector_indices.push_back(group_of[g]);
    S->group_starts.push_back(groups[group_of[g]].first);
    S->group_ends.push_back(groups[group_of[g]].second);
    S->program_offsets.push_back(-1);
  }
  auto attach = [&](int key, const Prog& p) {
    S->program_offsets[idx_of[key]] = (int32_t)S->programs.size();
    S->programs.insert(S->programs.end(), p.w.begin(), p.w.end());
  };
  if (arith_prog) attach(K_ARITH, program_arithmetic(20));
  if (with_prog) {
    attach(K_ARITHEXT, program_arithmetic_extension(10));
    attach(K_BASESUM, program_base_sum(63));
  }
  if (more_prog) {
    attach(K_EXP, program_exponentiation(66));
    attach(K_RANDACC, program_random_access(4, 4, 2));
  }
  if (rec_prog) {
    attach(K_MULEXT, program_mul_extension(13));
    attach(K_REDUCING, program_reducing(43, false));
    attach(K_REDUCINGEXT, program_reducing(32, true));
    attach(K_POSEIDONMDS, program_poseidon_mds());
    attach(K_COSETINTERP, program_coset_interpolation());
  }
  if (u32_prog) {
    attach(K_U32ARITH, program_u32_arithmetic(3));
    attach(K_U32ADDMANY, program_u32_add_many(3, 5));
    attach(K_U32SUB, program_u32_subtraction(6));
    attach(K_U32RANGE, program_u32_range_check(7));
    attach(K_COMPARISON, program_comparison(16));
  }

  // ---- row budget ----
  const size_t body = n - 3;
  size_t n_noop = std::max<size_t>(1, body / 64);
  if (n_noop > body) n_noop = body;
  size_t n_ext = with_prog ? std::max<size_t>(1, body / 16) : 0, n_bs = with_prog ? std::max<size_t>(1, body / 16) : 0;
  size_t n_exp = more_prog ? std::max<size_t>(1, body / 32) : 0, n_ra = more_prog ? std::max<size_t>(1, body / 32) : 0;
  size_t n_rec = rec_prog ? std::max<size_t>(1, body / 64) : 0;   // rows of EACH of the five recursion gates
  size_t n_recs[5] = {n_rec, n_rec, n_rec, n_rec, n_rec};         // MulExtension, Reducing, ReducingExtension, PoseidonMds, CosetInterpolation
  // lookup argument: a table of L (input, output) pairs in ceil(L / 26) LookupTableGate rows, n_lu LookupGate rows, and the
  // all-zero NoopGate row that must follow the table (circuit_builder.rs::add_all_lookups)
  const size_t lut_len = with_lookup ? std::min<size_t>(200, 26 * std::max<size_t>(1, body / 32)) : 0;
  size_t n_lut = with_lookup ? (lut_len + 25) / 26 : 0, n_lu = with_lookup ? std::max<size_t>(1, body / 32) : 0;
  // the U32 "voting threshold" block: n_u32 chained U32ArithmeticGate rows + 1 threshold row, and n_u32 rows of each of the
  // other four gates
  size_t n_u32 = u32_prog ? std::max<size_t>(1, body / 64) : 0;
  auto u32_total = [&]() { return n_u32 ? 5 * n_u32 + 1 : (size_t)0; };
  size_t n_pos = 0, n_arith = 0;
  if (mix) {
    // the declared mix: every enabled family gets its share of the trace (at least one row), the rest is padding
    for (int i = 0; i < VXS_MIX_COUNT; ++i)
      if (mix[i] < 0 || mix[i] > 1000) { delete S; return nullptr; }
    auto share = [&](int i, bool enabled) -> size_t {
      if (!enabled) return 0;
      return std::max<size_t>(1, (size_t)((unsigned __int128)body * (unsigned)mix[i] / 1000));
    };
    if ((!with_prog && (mix[VXS_MIX_ARITHMETIC_EXTENSION] || mix[VXS_MIX_BASE_SUM])) || (!more_prog && (mix[VXS_MIX_EXPONENTIATION] || mix[VXS_MIX_RANDOM_ACCESS])) ||
        (!rec_prog && (mix[VXS_MIX_MUL_EXTENSION] || mix[VXS_MIX_REDUCING] || mix[VXS_MIX_REDUCING_EXTENSION] || mix[VXS_MIX_POSEIDON_MDS] || mix[VXS_MIX_COSET_INTERPOLATION])) ||
        (!with_lookup && mix[VXS_MIX_LOOKUP]) || (!has_poseidon && mix[VXS_MIX_POSEIDON])) { delete S; return nullptr; }
    n_noop = 0;
    n_ext = share(VXS_MIX_ARITHMETIC_EXTENSION, with_prog), n_bs = share(VXS_MIX_BASE_SUM, with_prog);
    n_exp = share(VXS_MIX_EXPONENTIATION, more_prog), n_ra = share(VXS_MIX_RANDOM_ACCESS, more_prog);
    for (int q = 0; q < 5; ++q) n_recs[q] = share(VXS_MIX_MUL_EXTENSION + q, rec_prog);
    n_lu = share(VXS_MIX_LOOKUP, with_lookup);
    n_pos = has_poseidon ? (size_t)((unsigned __int128)body * (unsigned)mix[VXS_MIX_POSEIDON] / 1000) : 0;
    n_arith = (size_t)((unsigned __int128)body * (unsigned)mix[VXS_MIX_ARITHMETIC] / 1000);
  }
  auto rec_total = [&]() { return n_recs[0] + n_recs[1] + n_recs[2] + n_recs[3] + n_recs[4]; };
  const size_t lookup_total = (with_lookup ? n_lu + n_lut + 1 : 0);
  if (!mix)
    while (n_noop + n_ext + n_bs + n_exp + n_ra + rec_total() + lookup_total + u32_total() > body && n_ext + n_exp + (n_recs[0] > 1) + (n_u32 > 1) > 0) {
      if (n_ext) --n_ext, --n_bs;
      if (n_exp) --n_exp, --n_ra;
      if (n_recs[0] > 1) for (size_t& r : n_recs) --r;
      if (n_u32 > 1) --n_u32;
    }
  if (mix) {   // a tiny trace: the one-row minimum of every family comes out of the arithmetic rows first, then out of the Poseidon rows
    const size_t fixed = n_ext + n_bs + n_exp + n_ra + rec_total() + lookup_total + u32_total();
    if (fixed > body) { delete S; return nullptr; }
    if (fixed + n_pos > body) n_pos = body - fixed;
    if (fixed + n_pos + n_arith > body) n_arith = body - fixed - n_pos;
  }
  if (n_noop + n_ext + n_bs + n_exp + n_ra + rec_total() + lookup_total + u32_total() + n_pos + n_arith > body) { delete S; return nullptr; }
  S->n_u32 = n_u32;
  S->n_recursion = n_recs[0];
  S->n_lookup = n_lu;
  S->n_lookup_table = n_lut;
  size_t rest = body - n_noop - n_ext - n_bs - n_exp - n_ra - rec_total() - lookup_total - u32_total();
  S->n_exp = n_exp;
  S->n_randacc = n_ra;
  if (!mix) {
    n_pos = rest * (size_t)poseidon_percent / 100;
    n_arith = rest - n_pos;
  }
  S->n_poseidon = has_poseidon ? n_pos + 1 : 0;
  S->n_arith = n_arith;
  S->n_noop = n_noop;
  S->n_arithext = n_ext;
  S->n_basesum = n_bs;

  S->witness.assign((size_t)NW * n, 0);
  S->constants_sigmas.assign((size_t)(NCONST + NR) * n, 0);
  u64* w = S->witness.data();
  u64* c0 = &S->constants_sigmas[(size_t)(NSEL + NLS) * n];
  u64* c1 = &S->constants_sigmas[(size_t)(NSEL + NLS + 1) * n];
  const u64 UNUSED = 0xFFFFFFFFULL;
  auto set_gate = [&](size_t row, int key) {
    const int g = idx_of[key];
    ++S->rows_by_key[key];
    for (int q = 0; q < NSEL; ++q) S->constants_sigmas[(size_t)q * n + row] = group_of[g] == q ? (u64)g : UNUSED;
  };
  DSU dsu((size_t)NR * n);
  auto cell = [&](int col, size_t row) { return (uint32_t)((size_t)col * n + row); };

  // public inputs
  S->public_inputs.resize(4);
  for (auto& v : S->public_inputs) v = wrng.field();
  // row 1: constants 0, 1
  set_gate(1, K_CONST);
  c0[1] = 0;
  c1[1] = 1;
  w[0 * n + 1] = 0;
  w[1 * n + 1] = 1;
  // row 2: in-circuit hash of the public inputs
  set_gate(2, has_poseidon ? K_POSEIDON : K_NOOP);
  u64 in[12] = {0}, out[12];
  for (int i = 0; i < 4; ++i) in[i] = S->public_inputs[i];
  fill_poseidon_row(w, n, 2, in, out);
  for (int i = 0; i < 4; ++i) {
    S->pi_rows.push_back(2);
    S->pi_cols.push_back((uint32_t)i);
  }
  for (int j = 4; j < 12; ++j) dsu.unite(cell(j, 2), cell(0, 1));  // zero padding of the sponge
  dsu.unite(cell(24, 2), cell(0, 1));                              // swap = 0
  // row 0: public input gate carries the hash
  set_gate(0, K_PI);
  for (int i = 0; i < 4; ++i) {
    w[(size_t)i * n + 0] = out[i];
    dsu.unite(cell(i, 0), cell(12 + i, 2));
  }
  // body: interleave Poseidon and arithmetic rows so both gate kinds are spread over the trace
  size_t row = 3, pos_left = n_pos, ar_left = n_arith;
  size_t prev_pos_row = 0;
  bool have_prev_pos = false;  // the hash chain starts from free inputs, so only rows 0 and 2 depend on the public inputs
  bool have_prev_arith = false;
  size_t prev_arith_row = 0;
  u64 prev_arith_out = 0;
  while (pos_left + ar_left > 0) {
    bool do_pos = pos_left * (n_arith + 1) >= ar_left * (n_pos + 1) ? pos_left > 0 : false;
    if (!do_pos && ar_left == 0) do_pos = true;
    if (do_pos) {
      set_gate(row, K_POSEIDON);
      if (have_prev_pos) memcpy(in, out, sizeof in);
      else for (int i = 0; i < 12; ++i) in[i] = wrng.field();
      fill_poseidon_row(w, n, row, in, out);
      if (have_prev_pos) for (int i = 0; i < 12; ++i) dsu.unite(cell(i, row), cell(12 + i, prev_pos_row));
      have_prev_pos = true;
      prev_pos_row = row;
      --pos_left;
    } else {
      set_gate(row, K_ARITH);
      u64 k0 = rng.field(), k1 = rng.field();
      c0[row] = k0;
      c1[row] = k1;
      for (int op = 0; op < 20; ++op) {
        u64 m0 = (op == 0 && !have_prev_arith) ? wrng.field() : prev_arith_out;
        u64 m1 = wrng.field(), ad = wrng.field();
        u64 o = add(mul(mul(m0, m1), k0), mul(ad, k1));
        w[(size_t)(4 * op) * n + row] = m0;
        w[(size_t)(4 * op + 1) * n + row] = m1;
        w[(size_t)(4 * op + 2) * n + row] = ad;
        w[(size_t)(4 * op + 3) * n + row] = o;
        if (op > 0) dsu.unite(cell(4 * op, row), cell(4 * op - 1, row));
        else if (have_prev_arith) dsu.unite(cell(0, row), cell(79, prev_arith_row));
        prev_arith_out = o;
      }
      have_prev_arith = true;
      prev_arith_row = row;
      --ar_left;
    }
    ++row;
  }
  // ArithmeticExtensionGate rows: 10 ops in F_p^2, op k's first multiplicand copy-constrained to op k-1's output
  for (size_t e = 0; e < n_ext; ++e, ++row) {
    set_gate(row, K_ARITHEXT);
    const u64 k0 = rng.field(), k1 = rng.field();
    c0[row] = k0;
    c1[row] = k1;
    Ext prev{0, 0};
    for (int op = 0; op < 10; ++op) {
      Ext m0 = op == 0 ? Ext{wrng.field(), wrng.field()} : prev;
      Ext m1{wrng.field(), wrng.field()}, ad{wrng.field(), wrng.field()};
      Ext pr = emul(m0, m1);
      Ext o{add(mul(pr.a, k0), mul(ad.a, k1)), add(mul(pr.b, k0), mul(ad.b, k1))};
      const u64 vals[8] = {m0.a, m0.b, m1.a, m1.b, ad.a, ad.b, o.a, o.b};
      for (int k = 0; k < 8; ++k) w[(size_t)(8 * op + k) * n + row] = vals[k];
      if (op > 0) {
        dsu.unite(cell(8 * op, row), cell(8 * op - 2, row));
        dsu.unite(cell(8 * op + 1, row), cell(8 * op - 1, row));
      }
      prev = o;
    }
  }
  // BaseSumGate<2> rows: wire 0 = sum of 63 binary limbs (wires 1..63, little-endian)
  for (size_t b = 0; b < n_bs; ++b, ++row) {
    set_gate(row, K_BASESUM);
    const u64 v = wrng.next() >> 1;  // < 2^63 < p
    w[0 * n + row] = v;
    for (int i = 0; i < 63; ++i) w[(size_t)(1 + i) * n + row] = (v >> i) & 1;
  }
  // ExponentiationGate rows: output = base^(66-bit exponent), intermediates per gates/exponentiation.rs
  for (size_t e = 0; e < n_exp; ++e, ++row) {
    set_gate(row, K_EXP);
    const int nb = 66;
    const u64 base = wrng.field();
    w[0 * n + row] = base;
    int bits[66];
    for (int i = 0; i < nb; ++i) bits[i] = (int)(wrng.next() & 1), w[(size_t)(1 + i) * n + row] = (u64)bits[i];
    u64 acc = 0;
    for (int i = 0; i < nb; ++i) {
      u64 prev = i == 0 ? 1 : mul(acc, acc);
      int cur = bits[nb - i - 1];
      acc = mul(prev, cur ? base : 1);
      w[(size_t)(2 + nb + i) * n + row] = acc;
    }
    w[(size_t)(1 + nb) * n + row] = acc;
  }
  // RandomAccessGate rows: 4 copies of a 16-entry list lookup + 2 extra constants
  for (size_t e = 0; e < n_ra; ++e, ++row) {
    set_gate(row, K_RANDACC);
    const int bits = 4, vec = 16, copies = 4, per = 18, routed = per * copies + 2;
    const u64 k0 = rng.field(), k1 = rng.field();
    c0[row] = k0;
    c1[row] = k1;
    for (int c = 0; c < copies; ++c) {
      const int idx = (int)(wrng.next() & (vec - 1));
      u64 items[16];
      for (int i = 0; i < vec; ++i) items[i] = wrng.field(), w[(size_t)(per * c + 2 + i) * n + row] = items[i];
      w[(size_t)(per * c) * n + row] = (u64)idx;
      w[(size_t)(per * c + 1) * n + row] = items[idx];
      for (int i = 0; i < bits; ++i) w[(size_t)(routed + c * bits + i) * n + row] = (u64)((idx >> i) & 1);
    }
    w[(size_t)(per * copies) * n + row] = k0;
    w[(size_t)(per * copies + 1) * n + row] = k1;
  }
  if (rec_prog) {
    auto W = [&](int col, size_t r) -> u64& { return w[(size_t)col * n + r]; };
    auto rext = [&]() { return Ext{wrng.field(), wrng.field()}; };
    auto put = [&](int col, size_t r, Ext e) { W(col, r) = e.a, W(col + 1, r) = e.b; };
    // MulExtensionGate rows
    for (size_t e = 0; e < n_recs[0]; ++e, ++row) {
      set_gate(row, K_MULEXT);
      const u64 k0 = rng.field();
      c0[row] = k0;
      for (int i = 0; i < 13; ++i) {
        Ext m0 = rext(), m1 = rext(), pr = emul(m0, m1);
        put(6 * i, row, m0);
        put(6 * i + 2, row, m1);
        put(6 * i + 4, row, Ext{mul(pr.a, k0), mul(pr.b, k0)});
      }
    }
    // ReducingGate / ReducingExtensionGate rows: Horner accumulation of the coefficients at alpha
    for (int kind = 0; kind < 2; ++kind)
      for (size_t e = 0; e < n_recs[1 + kind]; ++e, ++row) {
        set_gate(row, kind ? K_REDUCINGEXT : K_REDUCING);
        const int nc = kind ? 32 : 43, cw = kind ? 2 : 1, start_accs = 6 + cw * nc;
        Ext alpha = rext(), acc = rext();
        put(2, row, alpha);
        put(4, row, acc);
        for (int i = 0; i < nc; ++i) {
          Ext coeff = kind ? rext() : Ext{wrng.field(), 0};
          W(6 + cw * i, row) = coeff.a;
          if (kind) W(6 + cw * i + 1, row) = coeff.b;
          acc = eadd(emul(acc, alpha), coeff);
          put(i == nc - 1 ? 0 : start_accs + 2 * i, row, acc);
        }
      }
    // PoseidonMdsGate rows
    for (size_t e = 0; e < n_recs[3]; ++e, ++row) {
      set_gate(row, K_POSEIDONMDS);
      static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
      Ext in[12];
      for (int i = 0; i < 12; ++i) in[i] = rext(), put(2 * i, row, in[i]);
      for (int r = 0; r < 12; ++r) {
        Ext o{0, 0};
        for (int i = 0; i < 12; ++i) {
          const Ext v = in[(i + r) % 12];
          o = eadd(o, Ext{mul(v.a, C[i]), mul(v.b, C[i])});
        }
        if (r == 0) o = eadd(o, Ext{mul(in[0].a, 8), mul(in[0].b, 8)});
        put(24 + 2 * r, row, o);
      }
    }
    // CosetInterpolationGate rows
    CosetInterp G;
    for (size_t e = 0; e < n_recs[4]; ++e, ++row) {
      set_gate(row, K_COSETINTERP);
      u64 shift = wrng.field();
      if (shift == 0) shift = 7;
      const Ext point = rext();
      const u64 sinv = inv(shift);
      const Ext z{mul(point.a, sinv), mul(point.b, sinv)};
      W(G.w_shift(), row) = shift;
      put(G.w_point(), row, point);
      put(G.w_shifted(), row, z);
      Ext eval{0, 0}, prod{1, 0};
      int j = 0;
      for (int c = 0; c <= G.NI; ++c) {
        for (; j < G.chunk_end[c]; ++j) {
          const Ext v = rext();
          put(G.w_value(j), row, v);
          const Ext zx{sub(z.a, G.domain[j]), z.b};
          const Ext wv{mul(v.a, G.weights[j]), mul(v.b, G.weights[j])};
          eval = eadd(emul(eval, zx), emul(prod, wv));
          prod = emul(prod, zx);
        }
        if (c < G.NI) put(G.w_ieval(c), row, eval), put(G.w_iprod(c), row, prod);
        else put(G.w_eval(), row, eval);
      }
    }
  }
  if (u32_prog) {
    // ---- the voting-threshold block (circuits/builder/justification.rs:164-186): num_signed = sum of the validators'
    // "signed" bits through chained U32 additions (U32ArithmeticGate m0 * 1 + bit), then num_signed * 3 and
    // num_active * 2 (two more ops of that gate) and a ComparisonGate on the two products ----
    auto W = [&](int col, size_t r) -> u64& { return w[(size_t)col * n + r]; };
    auto u32r = [&]() { return wrng.next() >> 32; };
    auto put_limbs = [&](int base, int count, u64 v, size_t r) {
      for (int j = 0; j < count; ++j) W(base + j, r) = (v >> (2 * j)) & 3;
    };
    const u64 U32MAX = 0xFFFFFFFFULL;
    auto fill_arith_op = [&](size_t r, int op, u64 m0, u64 m1, u64 ad) {
      const u64 full = m0 * m1 + ad;   // < 2^64: (2^32-1)^2 + 2^32 - 1
      const u64 lo = full & U32MAX, hi = full >> 32;
      W(6 * op, r) = m0, W(6 * op + 1, r) = m1, W(6 * op + 2, r) = ad, W(6 * op + 3, r) = lo, W(6 * op + 4, r) = hi;
      W(6 * op + 5, r) = hi == U32MAX ? 0 : inv(U32MAX - hi);
      put_limbs(18 + 32 * op, 32, full, r);
      return lo;
    };
    u64 num_signed = 0, num_active = 0;
    size_t prev_row = 0;
    int prev_op = -1;
    for (size_t e = 0; e < n_u32; ++e, ++row) {
      set_gate(row, K_U32ARITH);
      for (int op = 0; op < 3; ++op) {
        const u64 bit = (wrng.next() & 7) != 0;   // 7 of 8 validators signed
        const u64 prev = num_signed;
        num_signed = fill_arith_op(row, op, prev, 1, bit);
        ++num_active;
        dsu.unite(cell(6 * op + 1, row), cell(1, 1));                       // multiplicand 1 = the constant one
        if (prev_op < 0) dsu.unite(cell(6 * op, row), cell(0, 1));          // num_signed starts from the constant zero
        else dsu.unite(cell(6 * op, row), cell(6 * prev_op + 3, prev_row)); // ... and continues from the previous low half
        prev_row = row, prev_op = op;
      }
    }
    const size_t thr_row = row++;
    set_gate(thr_row, K_U32ARITH);
    const u64 scaled_signed = fill_arith_op(thr_row, 0, num_signed, 3, 0);
    dsu.unite(cell(0, thr_row), cell(6 * prev_op + 3, prev_row));
    const u64 scaled_threshold = fill_arith_op(thr_row, 1, num_active, 2, 0);
    fill_arith_op(thr_row, 2, u32r(), u32r(), u32r());                      // a full-size product: both halves in use
    auto fill_comparison = [&](size_t r, u64 a, u64 b) {
      const ComparisonLayout L{16};
      W(0, r) = a, W(1, r) = b;
      u64 so_far = 0;
      for (int c = 0; c < 16; ++c) {
        const u64 fc = (a >> (2 * c)) & 3, sc = (b >> (2 * c)) & 3;
        W(L.first(c), r) = fc, W(L.second(c), r) = sc;
        const u64 diff = sub(sc, fc);
        const u64 eq = fc == sc;
        W(L.eq_dummy(c), r) = eq ? 1 : inv(diff);
        W(L.chunks_equal(c), r) = eq;
        const u64 inter = eq ? so_far : 0;
        W(L.intermediate(c), r) = inter;
        so_far = add(inter, eq ? 0 : diff);
      }
      W(3, r) = so_far;                                  // most_significant_diff in {-3..3}
      const u64 shifted = add(4, so_far);                // 2^chunk_bits + diff in [1, 7]
      for (int b2 = 0; b2 < 3; ++b2) W(L.msd_bit(b2), r) = (shifted >> b2) & 1;
      W(2, r) = (shifted >> 2) & 1;                      // = [a <= b]
    };
    for (size_t e = 0; e < n_u32; ++e, ++row) {
      set_gate(row, K_COMPARISON);
      if (e == 0) {
        fill_comparison(row, scaled_threshold, scaled_signed);
        dsu.unite(cell(0, row), cell(6 * 1 + 3, thr_row));
        dsu.unite(cell(1, row), cell(6 * 0 + 3, thr_row));
        if (scaled_threshold <= scaled_signed) dsu.unite(cell(2, row), cell(1, 1));   // is_valid_num_signed == true
      } else {
        const u64 a = u32r();
        fill_comparison(row, a, (e & 3) == 3 ? a : u32r());
      }
    }
    for (size_t e = 0; e < n_u32; ++e, ++row) {          // U32AddManyGate{3 addends, 5 ops}
      set_gate(row, K_U32ADDMANY);
      for (int op = 0; op < 5; ++op) {
        const u64 a0 = u32r(), a1 = u32r(), a2 = u32r(), cin = wrng.next() & 3;
        const u64 sum = a0 + a1 + a2 + cin;
        W(6 * op, row) = a0, W(6 * op + 1, row) = a1, W(6 * op + 2, row) = a2, W(6 * op + 3, row) = cin;
        W(6 * op + 4, row) = sum & U32MAX, W(6 * op + 5, row) = sum >> 32;
        put_limbs(30 + 18 * op, 16, sum & U32MAX, row);
        put_limbs(30 + 18 * op + 16, 2, sum >> 32, row);
      }
    }
    for (size_t e = 0; e < n_u32; ++e, ++row) {          // U32SubtractionGate{6 ops}: a 192-bit subtraction, borrow chained
      set_gate(row, K_U32SUB);
      u64 borrow = 0;
      for (int op = 0; op < 6; ++op) {
        const u64 x = u32r(), y = u32r();
        const bool under = x < y + borrow;
        const u64 res = (x - y - borrow) & U32MAX;
        W(5 * op, row) = x, W(5 * op + 1, row) = y, W(5 * op + 2, row) = borrow, W(5 * op + 3, row) = res, W(5 * op + 4, row) = under;
        put_limbs(30 + 16 * op, 16, res, row);
        if (op > 0) dsu.unite(cell(5 * op + 2, row), cell(5 * (op - 1) + 4, row));
        else dsu.unite(cell(2, row), cell(0, 1));
        borrow = under;
      }
    }
    for (size_t e = 0; e < n_u32; ++e, ++row) {          // U32RangeCheckGate{7 limbs}
      set_gate(row, K_U32RANGE);
      for (int i = 0; i < 7; ++i) {
        const u64 v = u32r();
        W(i, row) = v;
        put_limbs(7 + 16 * i, 16, v, row);
      }
    }
  }
  if (with_lookup) {
    // rows [last_lu_row, last_lut_row) = LookupGate, [last_lut_row, first_lut_row] = LookupTableGate, first_lut_row + 1 = Noop (zeros)
    const size_t last_lu_row = row, last_lut_row = row + n_lu, first_lut_row = last_lut_row + n_lut - 1;
    for (size_t i = 0; i < lut_len; ++i) {
      S->lut_inputs.push_back((uint16_t)i);
      S->lut_outputs.push_back((uint16_t)((i * i * 31 + 7 * i + 3) & 0xFFFF));
    }
    S->lut_lens.push_back((int32_t)lut_len);
    S->lookup_rows = {(int32_t)last_lu_row, (int32_t)last_lut_row, (int32_t)first_lut_row};
    std::vector<u64> mult(lut_len, 0);
    for (size_t r = last_lu_row; r < last_lut_row; ++r) {
      set_gate(r, K_LOOKUP);
      for (int sl = 0; sl < 40; ++sl) {
        // the last slots of the last LookupGate row stay "unused": plonky2 fills them with the table's first entry
        const bool unused = r + 1 == last_lut_row && sl >= 33;
        const size_t e = unused ? 0 : (size_t)(wrng.next() % lut_len);
        w[(size_t)(2 * sl) * n + r] = S->lut_inputs[e];
        w[(size_t)(2 * sl + 1) * n + r] = S->lut_outputs[e];
        ++mult[e];
      }
    }
    for (size_t r = last_lut_row; r <= first_lut_row; ++r) {
      set_gate(r, K_LOOKUPTABLE);
      for (int sl = 0; sl < 26; ++sl) {
        const size_t e = (first_lut_row - r) * 26 + sl;   // LookupTableGenerator: entries run from first_lut_row downwards
        if (e < lut_len) {
          w[(size_t)(3 * sl) * n + r] = S->lut_inputs[e];
          w[(size_t)(3 * sl + 1) * n + r] = S->lut_outputs[e];
          w[(size_t)(3 * sl + 2) * n + r] = mult[e];
        }
      }
    }
    // lookup selectors: TransSre on the table rows, TransLdc on the looking rows, InitSre on the row after the table,
    // LastLdc on the last looking row; the table's "ends" selector on last_lut_row
    u64* ls = &S->constants_sigmas[(size_t)NSEL * n];
    for (size_t r = last_lut_row; r <= first_lut_row; ++r) ls[0 * n + r] = 1;
    for (size_t r = last_lu_row; r < last_lut_row; ++r) ls[1 * n + r] = 1;
    ls[2 * n + first_lut_row + 1] = 1;
    ls[3 * n + last_lu_row] = 1;
    ls[4 * n + last_lut_row] = 1;
    row = first_lut_row + 1;
    set_gate(row++, K_NOOP);
  }
  for (; row < n; ++row) set_gate(row, K_NOOP);
  S->n_noop = S->rows_by_key[K_NOOP];

  // k_is = 7^j (plonk_common / circuit_builder: get_unique_coset_shifts)
  S->k_is.resize(NR);
  {
    u64 acc = 1;
    for (int j = 0; j < NR; ++j) {
      S->k_is[j] = acc;
      acc = mul(acc, 7);
    }
  }
  // sigma: every copy class becomes one cycle over its cells in ascending (column-major) order
  {
    std::vector<u64> subgroup(n);
    u64 wg = root_of_unity(degree_bits);
    subgroup[0] = 1;
    for (size_t i = 1; i < n; ++i) subgroup[i] = mul(subgroup[i - 1], wg);
    const size_t ncells = (size_t)NR * n;
    std::vector<uint32_t> next(ncells);
    std::vector<uint32_t> last_of_root(ncells, 0xFFFFFFFFu), first_of_root(ncells, 0xFFFFFFFFu);
    for (size_t cidx = 0; cidx < ncells; ++cidx) {
      uint32_t r = dsu.find((uint32_t)cidx);
      if (first_of_root[r] == 0xFFFFFFFFu) first_of_root[r] = (uint32_t)cidx;
      else next[last_of_root[r]] = (uint32_t)cidx;
      last_of_root[r] = (uint32_t)cidx;
    }
    for (size_t cidx = 0; cidx < ncells; ++cidx) {
      uint32_t r = dsu.find((uint32_t)cidx);
      if (last_of_root[r] == cidx) next[cidx] = first_of_root[r];
    }
    u64* sig = &S->constants_sigmas[(size_t)NCONST * n];
    for (size_t cidx = 0; cidx < ncells; ++cidx) {
      size_t jc = next[cidx] / n, ir = next[cidx] % n;
      sig[cidx] = mul(S->k_is[jc], subgroup[ir]);
    }
  }
  vx_circuit_desc& d = S->desc;
  memset(&d, 0, sizeof d);
  d.degree_bits = degree_bits;
  d.num_wires = NW;
  d.num_routed_wires = NR;
  d.num_challenges = 2;
  d.rate_bits = 3;
  d.cap_height = std::min(4, degree_bits + 3);
  d.pow_bits = 16;
  d.num_query_rounds = 28;
  d.quotient_degree_factor = S->quotient_degree_factor;
  d.num_gates = ng;
  d.gate_types = S->gate_types.data();
  d.gate_params = S->gate_params.data();
  d.selector_indices = S->selector_indices.data();
  d.group_starts = S->group_starts.data();
  d.group_ends = S->group_ends.data();
  d.num_selectors = NSEL;
  d.num_constants = NCONST;
  d.constants_sigmas = S->constants_sigmas.data();
  d.k_is = S->k_is.data();
  d.num_public_inputs = 4;
  d.pi_rows = S->pi_rows.data();
  d.pi_cols = S->pi_cols.data();
  d.programs_len = (int32_t)S->programs.size();
  d.programs = S->programs.empty() ? nullptr : S->programs.data();
  d.program_offsets = S->program_offsets.data();
  if (with_lookup) {
    d.num_luts = 1;
    d.num_lookup_selectors = NLS;
    d.lut_lens = S->lut_lens.data();
    d.lut_inputs = S->lut_inputs.data();
    d.lut_outputs = S->lut_outputs.data();
    d.lookup_rows = S->lookup_rows.data();
  }
  return reinterpret_cast<vxs_circuit*>(S);
}

void vxs_free(vxs_circuit* c) { delete reinterpret_cast<Synth*>(c); }
/* New public inputs for the same circuit: only witness rows 0 (PublicInputGate) and 2 (in-circuit hash of the
 * public inputs) change.  Writes the two rows (num_wires values each) for the caller to patch into its copy of the
 * witness matrix, and updates the generator's own copy if it is still held. */
void vxs_patch_public_inputs(vxs_circuit* c, const uint64_t pi[4], uint64_t* row0_out, uint64_t* row2_out) {
  Synth* S = reinterpret_cast<Synth*>(c);
  const int NW = 135;
  std::vector<u64> tmp((size_t)NW * 4, 0);  // a 4-row scratch witness, rows 0 and 2 used
  u64 in[12] = {0}, out[12];
  for (int i = 0; i < 4; ++i) in[i] = canon(pi[i]);
  fill_poseidon_row(tmp.data(), 4, 2, in, out);
  for (int i = 0; i < 4; ++i) tmp[(size_t)i * 4 + 0] = out[i];
  for (int col = 0; col < NW; ++col) {
    row0_out[col] = tmp[(size_t)col * 4 + 0];
    row2_out[col] = tmp[(size_t)col * 4 + 2];
  }
  for (int i = 0; i < 4; ++i) S->public_inputs[i] = in[i];
  if (!S->witness.empty())
    for (int col = 0; col < NW; ++col) {
      S->witness[(size_t)col * S->n + 0] = row0_out[col];
      S->witness[(size_t)col * S->n + 2] = row2_out[col];
    }
}
const vx_circuit_desc* vxs_desc(vxs_circuit* c) { return &reinterpret_cast<Synth*>(c)->desc; }
const uint64_t* vxs_witness(vxs_circuit* c) { return reinterpret_cast<Synth*>(c)->witness.data(); }
const uint64_t* vxs_public_inputs(vxs_circuit* c) { return reinterpret_cast<Synth*>(c)->public_inputs.data(); }
void vxs_row_counts(vxs_circuit* c, uint64_t out[3]) {
  Synth* S = reinterpret_cast<Synth*>(c);
  out[0] = S->n_poseidon;
  out[1] = S->n_arith;
  out[2] = S->n_noop;
}
/* Rows per gate: out[key] for the VXS gate keys (vxs_gate_key_name), and the key of every gate of the description in its order. */
int vxs_gate_rows(vxs_circuit* c, uint64_t* out, int cap) {
  Synth* S = reinterpret_cast<Synth*>(c);
  for (int k = 0; k < K_COUNT && k < cap; ++k) out[k] = S->rows_by_key[k];
  return K_COUNT;
}
int vxs_gate_keys(vxs_circuit* c, int32_t* out, int cap) {
  Synth* S = reinterpret_cast<Synth*>(c);
  for (size_t g = 0; g < S->key_of_gate.size() && (int)g < cap; ++g) out[g] = S->key_of_gate[g];
  return (int)S->key_of_gate.size();
}
const char* vxs_gate_key_name(int key) {
  static const char* names[K_COUNT] = {"NoopGate", "ConstantGate", "PublicInputGate", "ArithmeticGate", "PoseidonGate", "ArithmeticExtensionGate",
                                       "BaseSumGate", "ExponentiationGate", "RandomAccessGate", "MulExtensionGate", "ReducingGate",
                                       "ReducingExtensionGate", "PoseidonMdsGate", "CosetInterpolationGate", "LookupGate", "LookupTableGate",
                                       "U32ArithmeticGate", "U32AddManyGate", "U32SubtractionGate", "U32RangeCheckGate", "ComparisonGate"};
  return key >= 0 && key < K_COUNT ? names[key] : "";
}
uint64_t vxs_recursion_rows(vxs_circuit* c) { return reinterpret_cast<Synth*>(c)->n_recursion; }
uint64_t vxs_u32_rows(vxs_circuit* c) { return reinterpret_cast<Synth*>(c)->n_u32; }
void vxs_row_counts_ext(vxs_circuit* c, uint64_t out[7]) {
  Synth* S = reinterpret_cast<Synth*>(c);
  out[0] = S->n_poseidon;
  out[1] = S->n_arith;
  out[2] = S->n_noop;
  out[3] = S->n_arithext;
  out[4] = S->n_basesum;
  out[5] = S->n_exp;
  out[6] = S->n_randacc;
}
/* Drop the (large) sigma/witness host copies once they have been handed to a prover. */
void vxs_release_host_buffers(vxs_circuit* c, int witness, int preprocessed) {
  Synth* S = reinterpret_cast<Synth*>(c);
  if (witness) std::vector<u64>().swap(S->witness);
  if (preprocessed) {
    std::vector<u64>().swap(S->constants_sigmas);
    S->desc.constants_sigmas = nullptr;
  }
}

}  // extern "C"

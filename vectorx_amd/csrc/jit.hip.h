// Constraint programs compiled to native gfx950 code when a circuit is built.
//
// A gate outside the native set arrives as a straight-line constraint program (include/vxprover.h VX_OP_*).  The
// interpreter (program_gates_kernel) keeps its 64 virtual registers in per-thread scratch memory and pays ~400 cycles
// per interpreted instruction per wavefront; a recursion circuit carries thousands of program words, which would cost
// more than the whole native quotient kernel.  So vx_circuit_create turns every program into HIP source — the same
// field primitives (goldilocks.hip.h, embedded at build time as jit_prelude.inc), one line per instruction, virtual
// registers as a local array with constant indices that the compiler promotes to VGPRs — compiles it with hiprtc for
// gfx950 and loads it as a code-object module.  ALL program gates of a circuit go into ONE kernel, one block of code
// per gate: evaluated gate by gate in separate launches the kernels are bound by re-reading the wire columns from HBM
// (each gate touches 60-130 of the 135 columns), fused they share one pass over the row.  Everything that differs
// between circuits using the same gate set (selector columns, group ranges, position of the constraints in the
// alpha-power table, ...) is a RUNTIME argument, so a gate set is compiled once per process.  hiprtc is reached
// through dlopen: when it is missing, or VX_NO_JIT=1, or the compile fails, the gates stay on the interpreter —
// still on the GPU, same values (tests/test_gpu_prover.py compares the two paths).
#pragma once
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hiprtc.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <atomic>
#include <sstream>
#include <thread>
#include <string>
#include <vector>

#define VX_AIR_MAX_PI 64          /* public inputs of an AIR (AirParams::pi, stark.hip.h) */
#define VX_AIR_MAX_CHALLENGES 16  /* second-round challenges of an AIR (AirParams::chal) */
// The generated sources mirror JitGateParams / AirParams textually: every array bound and the root-table shift are
// emitted from the SAME macros the host structs use (jit_limits_defines), so moving a limit cannot shift the
// kernel-argument layout of one side only.
static std::string jit_limits_defines() {
  std::ostringstream s;
  s << "#define VX_MAX_CHALLENGES " << VX_MAX_CHALLENGES << "\n#define VX_MAX_RATE " << VX_MAX_RATE << "\n#define VX_MAX_PROGRAM_GATES " << VX_MAX_PROGRAM_GATES
    << "\n#define VX_AIR_MAX_PI " << VX_AIR_MAX_PI << "\n#define VX_AIR_MAX_CHALLENGES " << VX_AIR_MAX_CHALLENGES << "\n#define VX_PROGRAM_REGS " << VX_PROGRAM_REGS
    << "\n#define ROOT_TABLE_LOG " << ROOT_TABLE_LOG << "\n";
  return s.str();
}

#define VX_JIT_GROUP_MAX 8   /* program gates per generated kernel (JitGateParams::g); emitted into the generated source by jit_limits_defines */
struct JitGateRt {
  int gate_index, selector_index, group_start, group_end;
};
struct JitGateParams {  // mirrored textually in jit_gate_source()
  const u64 *cs, *wires;
  const u64* alpha_pows;
  const Limbs3x2* alpha_limbs;   // the same powers pre-split for carry-free accumulation (poseidon.hip.h dot3)
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch;
  int base_idx, ngates;
  int const_base, pad_;  // first gate constant among the preprocessed columns: num_selectors + num_lookup_selectors
  u64 pih[4];
  u64 zh_inv[VX_MAX_RATE];
  JitGateRt g[VX_JIT_GROUP_MAX];  // the gates this launch evaluates (a group of program gates per kernel)
};
static_assert(sizeof(JitGateParams::g) / sizeof(JitGateRt) == VX_JIT_GROUP_MAX, "host JitGateParams and the generated source share VX_JIT_GROUP_MAX");

static const char* JIT_PRELUDE =
#include "jit_prelude.inc"
    ;

struct JitApi {
  void* handle = nullptr;
  hiprtcResult (*create)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  hiprtcResult (*compile)(hiprtcProgram, int, const char**) = nullptr;
  hiprtcResult (*log_size)(hiprtcProgram, size_t*) = nullptr;
  hiprtcResult (*log)(hiprtcProgram, char*) = nullptr;
  hiprtcResult (*code_size)(hiprtcProgram, size_t*) = nullptr;
  hiprtcResult (*code)(hiprtcProgram, char*) = nullptr;
  hiprtcResult (*destroy)(hiprtcProgram*) = nullptr;
  hiprtcResult (*version)(int*, int*) = nullptr;  // optional (cache key only)
  bool ok = false;
};
static JitApi& jit_api() {
  static JitApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) return;
#define VX_JIT_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, name)
    VX_JIT_SYM(create, "hiprtcCreateProgram");
    VX_JIT_SYM(compile, "hiprtcCompileProgram");
    VX_JIT_SYM(log_size, "hiprtcGetProgramLogSize");
    VX_JIT_SYM(log, "hiprtcGetProgramLog");
    VX_JIT_SYM(code_size, "hiprtcGetCodeSize");
    VX_JIT_SYM(code, "hiprtcGetCode");
    VX_JIT_SYM(destroy, "hiprtcDestroyProgram");
    VX_JIT_SYM(version, "hiprtcVersion");
#undef VX_JIT_SYM
    api.ok = api.create && api.compile && api.log_size && api.log && api.code_size && api.code && api.destroy;
  });
  return api;
}

// Straight-line HIP source of one constraint program (the words up to VX_OP_END; already validated by the caller).
//   air == false: a gate program — LDW / LDC / LDP read the row's wires, the gate constants and the public-input hash;
//                 PUSH adds R[a] * alpha^k (k = position of the constraint) to a0 / a1.
//   air == true:  an AIR program (stark.hip.h) — LDW / LDN read the local / next trace row, LDP a public input; PUSH
//                 multiplies by the kind's factor (VX_AIR_*) and folds Horner-style, acc = acc * alpha + t.
struct JitIns { int op, dst, a, b; uint64_t imm; bool skip; int fa, fb; };  // fa/fb: multiplicands folded into an ADD
static std::vector<JitIns> jit_decode(const uint64_t* prog) {
  std::vector<JitIns> code;
  for (int pc = 0;; ++pc) {
    const uint64_t ins = prog[pc];
    JitIns I{(int)(ins & 0xFF), (int)((ins >> 8) & 63), (int)((ins >> 16) & 0xFFFF), (int)((ins >> 32) & 0xFFFF), 0, false, -1, -1};
    if (I.op == VX_OP_END) break;
    if (I.op == VX_OP_PUSH) I.b = (int)((ins >> 32) & 0xFFFF);  // AIR: the constraint kind
    if (I.op == VX_OP_LDI) I.imm = vxh::canon(prog[++pc]);
    if (I.op == VX_OP_ADD || I.op == VX_OP_SUB || I.op == VX_OP_MUL) I.a &= 63, I.b &= 63;
    if (I.op == VX_OP_PUSH) I.a &= 63;
    code.push_back(I);
  }
  return code;
}
// aux mode (jit_aux_source): PUSHes are (numerator, denominator) pairs of fractions, inverted in batches of VX_AUX_JIT_BATCH
#define VX_AUX_JIT_BATCH 8
struct JitAuxMode {
  bool on = false;
  int nfrac = 0;
};
static thread_local JitAuxMode g_jit_aux;
static thread_local bool g_jit_lds_wires = false;   // fused gate kernel: LDW reads the row's wires from the workgroup's LDS tile (already canonical)
static void jit_emit_code(std::ostringstream& s, std::vector<JitIns> code, int nch, bool air, int air_ncols, int k0 = 0);
static void jit_emit_program(std::ostringstream& s, const uint64_t* prog, int nch, bool air, int air_ncols = 0) {
  jit_emit_code(s, jit_decode(prog), nch, air, air_ncols);
}
// `code`: a self-contained straight-line sequence (every register is written before it is read)
// `k0`: position (among its gate's constraints) of the first constraint `code` pushes — a CHUNK of a gate's program (jit_fused_plan)
static void jit_emit_code(std::ostringstream& s, std::vector<JitIns> code, int nch, bool air, int air_ncols, int k0) {
  typedef JitIns Ins;
  // AIR: columns >= air_ncols are second-round (aux) columns, held in their own LDE with the same row stride
  auto col = [&](int a) {
    std::ostringstream e;
    if (air && a >= air_ncols) e << "AUX[(size_t)" << (a - air_ncols) << " * SW + ";
    else e << "W[(size_t)" << a << " * SW + ";
    return e.str();
  };
  // ---- two straight-line optimisations before emitting ----------------------------------------
  //  (1) multiply-add fusion: a MUL whose result is read exactly once, by an ADD, becomes one gl_mad at the ADD
  //      (the F_p^2 products the emitters produce are chains of exactly this shape);
  //  (2) lazy canonicalisation: products stay arbitrary u64 representatives (gl_mul_nc / gl_mad_nc accept and return
  //      them, PUSH accepts them); a register is canonicalised only when an ADD / SUB is about to read it.
  auto reads = [](const Ins& I, int r) {
    if (I.op == VX_OP_ADD || I.op == VX_OP_SUB || I.op == VX_OP_MUL) return (I.a == r) + (I.b == r);
    if (I.op == VX_OP_PUSH) return (int)(I.a == r);
    return 0;
  };
  auto writes = [](const Ins& I, int r) { return I.op != VX_OP_PUSH && I.dst == r; };
  for (size_t i = 0; i < code.size(); ++i) {
    if (code[i].op != VX_OP_MUL) continue;
    const int d = code[i].dst, ma = code[i].a, mb = code[i].b;
    if (ma == d || mb == d) continue;  // the product overwrites one of its own factors: cannot be re-materialised later
    // the product must be read exactly once before d is overwritten (or the program ends), by an ADD, with the
    // multiplicands untouched in between
    int uses = 0;
    size_t use_at = 0;
    bool operands_live = true, ok = true;
    for (size_t j = i + 1; j < code.size(); ++j) {
      const int rd = reads(code[j], d);
      if (rd) {
        uses += rd;
        if (uses == 1) {
          use_at = j;
          ok = operands_live && code[j].op == VX_OP_ADD && code[j].fa < 0 && code[j].a != code[j].b;
        }
      }
      if (writes(code[j], d)) break;
      if (writes(code[j], ma) || writes(code[j], mb)) operands_live = false;
    }
    if (uses != 1 || !ok) continue;
    Ins& A = code[use_at];
    const int other = A.a == d ? A.b : A.a;
    A.fa = ma, A.fb = mb;
    A.a = other;  // the addend
    code[i].skip = true;
  }
  bool canon_reg[VX_PROGRAM_REGS];
  for (bool& c : canon_reg) c = true;
  // registers known to hold a small immediate (LDI of a value < 2^31, not overwritten since): a multiplication by one of them is
  // gl_mulc_nc / gl_madc_nc — 5-6 instructions instead of 16-18 (the 7 of every F_p^2 product, the MDS rows, the 2 of a base sum)
  bool imm_known[VX_PROGRAM_REGS];
  uint64_t imm_val[VX_PROGRAM_REGS];
  for (bool& c : imm_known) c = false;
  const bool small_consts = !getenv("VX_JIT_NO_SMALL_CONSTS");
  auto small = [&](int r) { return small_consts && imm_known[r] && imm_val[r] < (1ull << 31); };
  auto need_canon = [&](int r) {
    if (!canon_reg[r]) {
      s << "  R[" << r << "] = gl_canon(R[" << r << "]);\n";
      canon_reg[r] = true;
    }
  };
  int k = k0;
  for (const Ins& I : code) {
    if (I.skip) continue;
    if (I.op != VX_OP_PUSH && I.op != VX_OP_LDI) imm_known[I.dst] = false;
    switch (I.op) {
      case VX_OP_LDW:
        if (g_jit_lds_wires && !air) s << "  R[" << I.dst << "] = LW[" << I.a << " * 64 + lane];\n";
        else s << "  R[" << I.dst << "] = gl_canon(" << col(I.a) << "il]);\n";
        canon_reg[I.dst] = true;
        break;
      case VX_OP_LDN: s << "  R[" << I.dst << "] = gl_canon(" << col(I.a) << "il_next]);\n"; canon_reg[I.dst] = true; break;
      case VX_OP_LDCH: s << "  R[" << I.dst << "] = p.chal[" << I.a << "];\n"; canon_reg[I.dst] = true; break;
      case VX_OP_LDC: s << "  R[" << I.dst << "] = CS[(size_t)(cbase + " << I.a << ") * N + i];\n"; canon_reg[I.dst] = true; break;
      case VX_OP_LDI:
        s << "  R[" << I.dst << "] = " << I.imm << "ULL;\n";
        canon_reg[I.dst] = true;
        imm_known[I.dst] = true, imm_val[I.dst] = I.imm;
        break;
      case VX_OP_ADD:
        if (I.fa >= 0) {  // fused multiply-add: any representatives in, one out
          if (small(I.fb)) s << "  R[" << I.dst << "] = gl_madc_nc(R[" << I.fa << "], " << imm_val[I.fb] << "u, R[" << I.a << "]);\n";
          else if (small(I.fa)) s << "  R[" << I.dst << "] = gl_madc_nc(R[" << I.fb << "], " << imm_val[I.fa] << "u, R[" << I.a << "]);\n";
          else s << "  R[" << I.dst << "] = gl_mad_nc(R[" << I.fa << "], R[" << I.fb << "], R[" << I.a << "]);\n";
          canon_reg[I.dst] = false;
        } else {
          need_canon(I.a);
          need_canon(I.b);
          s << "  R[" << I.dst << "] = gl_add(R[" << I.a << "], R[" << I.b << "]);\n";
          canon_reg[I.dst] = true;
        }
        break;
      case VX_OP_SUB:
        need_canon(I.a);
        need_canon(I.b);
        s << "  R[" << I.dst << "] = gl_sub(R[" << I.a << "], R[" << I.b << "]);\n";
        canon_reg[I.dst] = true;
        break;
      case VX_OP_MUL:
        if (small(I.b)) s << "  R[" << I.dst << "] = gl_mulc_nc(R[" << I.a << "], " << imm_val[I.b] << "u);\n";
        else if (small(I.a)) s << "  R[" << I.dst << "] = gl_mulc_nc(R[" << I.b << "], " << imm_val[I.a] << "u);\n";
        else s << "  R[" << I.dst << "] = gl_mul_nc(R[" << I.a << "], R[" << I.b << "]);\n";
        canon_reg[I.dst] = false;
        break;
      case VX_OP_PUSH:
        if (g_jit_aux.on) {
          // fraction f = k / 2: numerator (any representative) then denominator (canonical: tested against zero); after every
          // VX_AUX_JIT_BATCH denominators — or the last one — the batch is inverted with ONE Fermat inversion (Montgomery's trick)
          const int f = k / 2;
          if ((k & 1) == 0) {
            s << "  const u64 num" << f << " = R[" << I.a << "];\n";
          } else {
            need_canon(I.a);
            s << "  const u64 den" << f << " = R[" << I.a << "];\n";
            const bool last = f + 1 == g_jit_aux.nfrac;
            if ((f + 1) % VX_AUX_JIT_BATCH == 0 || last) {
              const int f0 = f - (f % VX_AUX_JIT_BATCH);
              s << "  {\n    u64 acc = 1;\n";
              for (int j = f0; j <= f; ++j) s << "    const u64 z" << j << " = den" << j << " ? den" << j << " : 1; const u64 pre" << j << " = acc; acc = gl_mul(acc, z" << j << ");\n";
              s << "    u64 inv = aux_inv(acc);\n";
              for (int j = f; j >= f0; --j)
                s << "    { const u64 di = gl_mul(inv, pre" << j << "); inv = gl_mul(inv, z" << j << "); p.out[(size_t)p.frac_out[" << j << "] * p.n + il] = den" << j
                  << " ? gl_mul(num" << j << ", di) : 0; }\n";
              s << "  }\n";
            }
          }
        } else if (air) {
          const char* factor = I.b == VX_AIR_TRANSITION ? "z_last" : I.b == VX_AIR_FIRST_ROW ? "l_first" : I.b == VX_AIR_LAST_ROW ? "l_last" : nullptr;
          if (factor) s << "  { const u64 t = gl_mul_nc(R[" << I.a << "], " << factor << ");\n";
          else s << "  { const u64 t = R[" << I.a << "];\n";
          s << "    a0 = gl_mad(a0, p.alphas[0], t);\n";
          if (nch > 1) s << "    a1 = gl_mad(a1, p.alphas[1], t);\n";
          s << "  }\n";
        } else {
          // carry-free accumulation against the pre-split alpha powers (12 multiply-adds for both challenges instead of two
          // 18-instruction fused multiply-add-reduces): the native quotient kernel's acc_push
          s << "  dot3_mac(A0, R[" << I.a << "], AL[" << k << "]);\n";
          if (nch > 1) s << "  dot3_mac(A1, R[" << I.a << "], AL[" << (VX_ALPHA_POWS + k) << "]);\n";
        }
        ++k;
        break;
      case VX_OP_LDP:
        if (air) s << "  R[" << I.dst << "] = p.pi[" << I.a << "];\n";
        else s << "  R[" << I.dst << "] = p.pih[" << (I.a & 3) << "];\n";
        canon_reg[I.dst] = true;
        break;
      default: break;
    }
  }
}

// HIP source of the kernel of ONE program gate (round 3: one kernel per gate — the fused kernel of rounds 1-2 kept values
// of several gates live at once and spilled 1 kB per lane; a gate on its own gets the whole register file, compiles in a
// fraction of the time, and is cached per PROGRAM, so circuits that share gates share code objects).
static const char* JIT_DOT3 = R"VXJIT(
struct Limbs3x2 { u32 lo[3]; u32 hi[3]; };
struct dot3 { u64 s0, s1, s2; };
GLD void dot3_mac(dot3& D, u64 a, const Limbs3x2& b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
  D.s0 += (u64)a0 * b.lo[0];
  D.s1 += (u64)a0 * b.lo[1];
  D.s2 += (u64)a0 * b.lo[2];
  D.s0 += (u64)a1 * b.hi[0];
  D.s1 += (u64)a1 * b.hi[1];
  D.s2 += (u64)a1 * b.hi[2];
}
GLD u64 dot3_reduce_nc(const dot3& D) {
  typedef unsigned __int128 u128;
  const u128 V = (u128)D.s0 + ((u128)D.s1 << 22) + ((u128)D.s2 << 44);
  return gl_reduce128_nc((u64)V, (u64)(V >> 64));
}
)VXJIT";
static std::string jit_gate_source(const std::vector<const uint64_t*>& progs, int nch) {
  std::ostringstream s;
  // Occupancy bound: without one the compiler keeps every wire it has loaded live, takes >256 VGPRs and runs one wave per SIMD
  // (measured in round 1: 15.9 ms instead of 6.7); 4 blocks per CU = <= 128 VGPRs, like the native gate kernel.
  const char* bpc = getenv("VX_JIT_BLOCKS_PER_CU");
  s << "typedef unsigned long long uint64_t;\ntypedef unsigned int uint32_t;\n#define VX_JIT_BLOCKS_PER_CU " << (bpc ? atoi(bpc) : 4) << "\n"
    << "#define VX_ALPHA_POWS " << VX_ALPHA_POWS << "\n#define VX_JIT_GROUP_MAX " << VX_JIT_GROUP_MAX << "\n"
    << jit_limits_defines() << JIT_PRELUDE << JIT_DOT3 << R"VXJIT(
#define VX_JIT_GROUP_MAX 8   /* program gates per generated kernel (JitGateParams::g); emitted into the generated source by jit_limits_defines */
struct JitGateRt {
  int gate_index, selector_index, group_start, group_end;
};
struct JitGateParams {
  const u64 *cs, *wires;
  const u64* alpha_pows;
  const Limbs3x2* alpha_limbs;
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch;
  int base_idx, ngates;
  int const_base, pad_;
  u64 pih[4];
  u64 zh_inv[VX_MAX_RATE];
  JitGateRt g[VX_JIT_GROUP_MAX];
};
extern "C" __global__ __launch_bounds__(256, VX_JIT_BLOCKS_PER_CU) void vx_program_gate(JitGateParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const u32 z = (u32)(i >> p.log_n);
  const u32 r = p.rate_bits ? (__brev(z) >> (32 - p.rate_bits)) : 0u;
  const u64* __restrict__ CS = p.cs;
  const u64* __restrict__ W = p.wires;
  const Limbs3x2* __restrict__ AL = p.alpha_limbs + p.base_idx;
  const int nsel = p.num_selectors, cbase = p.const_base;
  u64 t0 = 0, t1 = 0;
)VXJIT";
  for (size_t q = 0; q < progs.size(); ++q) {
    s << "  {  // program gate, slot " << q << "\n"
         "    const JitGateRt G = p.g[" << q << "];\n"
         "    const u64 s = CS[(size_t)G.selector_index * N + i];\n"
         "    u64 filter = 1;\n"
         "    for (int q = G.group_start; q < G.group_end; ++q)\n"
         "      if (q != G.gate_index) filter = gl_mul(filter, gl_sub((u64)q, s));\n"
         "    if (nsel > 1) filter = gl_mul(filter, gl_sub(0xFFFFFFFFULL, s));\n"
         "    dot3 A0 = {0, 0, 0}, A1 = {0, 0, 0};\n"
         "    u64 R[VX_PROGRAM_REGS];\n";
    jit_emit_program(s, progs[q], nch, false);
    s << "    t0 = gl_mad(filter, dot3_reduce_nc(A0), t0);\n";
    if (nch > 1) s << "    t1 = gl_mad(filter, dot3_reduce_nc(A1), t1);\n";
    s << "    (void)A1;\n  }\n";
  }
  s << "  const u64 zi = p.zh_inv[r];\n"
       "  { u64* o = p.out + il; *o = gl_add(*o, gl_mul(t0, zi)); }\n";
  if (nch > 1) s << "  { u64* o = p.out + SW + il; *o = gl_add(*o, gl_mul(t1, zi)); }\n";
  s << "  (void)t1; (void)cbase;\n}\n";
  return s.str();
}
// How the program gates of a circuit are packed into kernels.  Measured at n = 2^20 with 9 program gates (flags 29, two runs
// each, gpurun_out r03g): ONE GATE PER KERNEL 13.4 ms  <  groups of <= 1500 instructions 16.1 - 16.6  <  everything in one kernel
// 18.2  <  groups of <= 3000 20.4 - 20.8 (and the all-in-one kernel of rounds 1-2 without the carry-free accumulation: 14.1 with
// 1 kB of scratch per lane).  Sharing the wire loads between gates does not pay for the registers it costs, so the default is one
// gate per kernel — which also makes the code-object cache per PROGRAM; VX_JIT_GATE_GROUP_INS > 1 packs consecutive gates up to
// that many instructions (at most VX_JIT_GROUP_MAX gates) for experiments.
static std::vector<std::vector<size_t>> jit_gate_groups(const std::vector<const uint64_t*>& progs) {
  const char* env = getenv("VX_JIT_GATE_GROUP_INS");
  const size_t budget = env && atoi(env) > 0 ? (size_t)atoi(env) : 1;
  std::vector<std::vector<size_t>> groups;
  size_t cur = 0;
  for (size_t q = 0; q < progs.size(); ++q) {
    size_t len = 0;
    for (int pc = 0; (progs[q][pc] & 0xFF) != VX_OP_END; ++pc) {
      ++len;
      if ((progs[q][pc] & 0xFF) == VX_OP_LDI) ++pc;
    }
    if (groups.empty() || cur + len > budget || groups.back().size() >= VX_JIT_GROUP_MAX) {
      groups.emplace_back();
      cur = 0;
    }
    groups.back().push_back(q);
    cur += len;
  }
  return groups;
}

// ---- ALL program gates of a circuit in ONE kernel that reads the wires ONCE (round 6) ---------------------------------------------
// One kernel per gate is bound by re-reading the wire columns: with the recursive verifier's nine program gates the wires LDE
// (135 columns x 8n rows) is read 813 columns' worth per proof, six of the nine kernels run at 4.2 - 4.7 TB/s of HBM, and the
// register-fused kernel of rounds 1-3 lost more to spills than it saved.  Here a workgroup owns 64 consecutive LDE rows and is W
// wavefronts wide: all waves stage the rows' wires into LDS once (column-major [column][64 rows]: a wave's 64 lanes write and read 512
// contiguous bytes, conflict-free; canonicalised on the way in), then every WAVE evaluates its own share of the gates on the same 64
// rows — a wave-uniform branch, so each gate's straight-line block keeps the register file to itself — reading wires from LDS at the
// point of use, and the waves' filtered sums meet in LDS for one read-modify-write of the quotient values.  Gates are dealt to waves
// by estimated instruction count (longest first).  HBM traffic of the program gates: the wires once.
#define VX_JIT_FUSED_MAX 24    /* program gates per fused kernel (JitFusedParams::g) */
#define VX_JIT_FUSED_WAVES 8   /* waves per workgroup: 8 x 64 threads, 2 workgroups per CU at 69 KB of LDS each = 4 waves per SIMD, 128 VGPRs */
struct JitFusedParams {  // mirrored textually in jit_fused_source()
  const u64 *cs, *wires;
  const Limbs3x2* alpha_limbs;
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch;
  int base_idx, ngates;
  int const_base, pad_;
  u64 pih[4];
  u64 zh_inv[VX_MAX_RATE];
  JitGateRt g[VX_JIT_FUSED_MAX];
};
struct JitFusedItem {   // a run of consecutive constraints of ONE gate: the code that computes them (backward slice of what came before + the run)
  size_t gate;          // index into `progs`
  int push_begin;       // position of its first constraint among the gate's constraints
  size_t cost;          // estimated VALU instructions
  std::vector<JitIns> code;
};
struct JitFusedPlan {
  int waves = 0, max_col = 0;
  std::vector<JitFusedItem> items;
  std::vector<std::vector<size_t>> of_wave;   // item indices per wave, grouped by gate
};
static size_t jit_cost_of(const std::vector<JitIns>& code, int nch) {
  size_t c = 0;
  for (const JitIns& I : code) {
    switch (I.op) {
      case VX_OP_MUL: c += 14; break;
      case VX_OP_ADD: case VX_OP_SUB: c += 5; break;
      case VX_OP_PUSH: c += 7 * (size_t)nch; break;
      default: c += 1; break;
    }
  }
  return c;
}
// instructions [begin, end) of a straight-line program preceded by the backward slice of everything they read but do not write
static std::vector<JitIns> jit_slice(const std::vector<JitIns>& code, size_t begin, size_t end) {
  auto srcs = [](const JitIns& I, int out[2]) {
    int k = 0;
    if (I.op == VX_OP_ADD || I.op == VX_OP_SUB || I.op == VX_OP_MUL) out[k++] = I.a, out[k++] = I.b;
    else if (I.op == VX_OP_PUSH) out[k++] = I.a;
    return k;
  };
  bool written[VX_PROGRAM_REGS] = {false}, need[VX_PROGRAM_REGS] = {false};
  for (size_t i = begin; i < end; ++i) {
    int sr[2];
    const int k = srcs(code[i], sr);
    for (int q = 0; q < k; ++q)
      if (!written[sr[q]]) need[sr[q]] = true;
    if (code[i].op != VX_OP_PUSH) written[code[i].dst] = true;
  }
  std::vector<char> take(begin, 0);
  for (size_t i = begin; i-- > 0;) {
    const JitIns& I = code[i];
    if (I.op == VX_OP_PUSH || !need[I.dst]) continue;
    take[i] = 1;
    need[I.dst] = false;
    int sr[2];
    const int k = srcs(I, sr);
    for (int q = 0; q < k; ++q) need[sr[q]] = true;
  }
  std::vector<JitIns> sub;
  for (size_t i = 0; i < begin; ++i)
    if (take[i]) sub.push_back(code[i]);
  sub.insert(sub.end(), code.begin() + begin, code.begin() + end);
  return sub;
}
// Work items = runs of constraints.  Nine gates do not balance on eight waves (the workgroup lasts as long as its longest wave: 5.5
// cycles per instruction against the issue floor's 4, profiles/r06_pmc_sq_prove_recursion.md), so a gate's program is cut at constraint
// boundaries into chunks of about two thirds of a wave's fair share; a chunk re-creates the registers it reads with the backward slice of
// the earlier code (the gates' chains — Reducing, Exponentiation, CosetInterpolation — restart from WIRES every step, so slices are a few
// loads), pushes against its own alpha powers (k0), and the chunks are dealt longest first.  A sum of filtered partial sums is the same
// field element as the filtered sum: byte-identical proofs (tests/test_gpu_prover.py).
static JitFusedPlan jit_fused_plan(const std::vector<const uint64_t*>& progs, int nch) {
  JitFusedPlan P;
  const char* we = getenv("VX_JIT_FUSED_WAVES");
  P.waves = std::max(1, std::min<int>({we && atoi(we) > 0 ? atoi(we) : VX_JIT_FUSED_WAVES, 16, (int)progs.size()}));
  std::vector<std::vector<JitIns>> codes;
  size_t total = 0;
  for (const uint64_t* prog : progs) {
    codes.push_back(jit_decode(prog));
    for (const JitIns& I : codes.back())
      if (I.op == VX_OP_LDW) P.max_col = std::max(P.max_col, I.a + 1);
    total += jit_cost_of(codes.back(), nch);
  }
  // VX_JIT_FUSED_CHUNK: 0 = whole gates; default = two thirds of a wave's fair share.  Measured at 2^18 rows, nine recursion gates, same
  // box: whole gates 2.59 ms (38 spilled VGPRs: the longest gate's block), a third of a share 2.80 (24 chunks: their slices and filters
  // cost more than the balance returns), two thirds 2.53 (no spills), a whole share 2.74 — balance is NOT what holds this kernel at 5.5
  // cycles per instruction; the chunks stay for the spills they remove.
  const char* ce = getenv("VX_JIT_FUSED_CHUNK");
  const size_t limit = ce ? (size_t)atoi(ce) : std::max<size_t>(400, 2 * total / ((size_t)P.waves * 3));
  for (size_t g = 0; g < codes.size(); ++g) {
    const std::vector<JitIns>& code = codes[g];
    size_t begin = 0;
    int pushes = 0;
    while (begin < code.size()) {
      size_t end = begin, cost = 0;
      int np = 0;
      while (end < code.size()) {
        const bool push = code[end].op == VX_OP_PUSH;
        cost += code[end].op == VX_OP_MUL ? 14 : (code[end].op == VX_OP_ADD || code[end].op == VX_OP_SUB) ? 5 : push ? 7 * (size_t)nch : 1;
        ++end;
        if (push) {
          ++np;
          if (limit && cost >= limit) break;
        }
      }
      if (end < code.size()) {   // never leave a tail without constraints
        bool more = false;
        for (size_t i = end; i < code.size(); ++i) more = more || code[i].op == VX_OP_PUSH;
        if (!more) end = code.size();
      }
      JitFusedItem it;
      it.gate = g, it.push_begin = pushes;
      it.code = jit_slice(code, begin, end);
      it.cost = jit_cost_of(it.code, nch);
      P.items.push_back(std::move(it));
      pushes += np;
      begin = end;
    }
  }
  P.of_wave.resize(P.waves);
  std::vector<size_t> order(P.items.size()), load(P.waves, 0);
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return P.items[a].cost > P.items[b].cost; });
  std::vector<std::vector<char>> has(P.waves, std::vector<char>(progs.size(), 0));
  for (size_t q : order) {
    // the least loaded wave; a wave that already holds a chunk of this gate saves the selector filter (150): prefer it on a near tie
    size_t best = 0, best_load = (size_t)-1;
    for (int w = 0; w < P.waves; ++w) {
      const size_t eff = load[w] + (has[w][P.items[q].gate] ? 0 : 150);
      if (eff < best_load) best_load = eff, best = (size_t)w;
    }
    P.of_wave[best].push_back(q);
    load[best] = best_load + P.items[q].cost;
    has[best][P.items[q].gate] = 1;
  }
  for (auto& v : P.of_wave)   // chunks of one gate next to each other, in program order: one filter, one pair of accumulators per gate and wave
    std::stable_sort(v.begin(), v.end(), [&](size_t a, size_t b) {
      return P.items[a].gate != P.items[b].gate ? P.items[a].gate < P.items[b].gate : P.items[a].push_begin < P.items[b].push_begin;
    });
  return P;
}
// can this gate set run as one fused kernel?  (at least two gates, no more than the parameter block holds, a wire tile that leaves room
// for two workgroups per CU)
static bool jit_fused_applicable(const std::vector<const uint64_t*>& progs, int nch) {
  const char* e = getenv("VX_JIT_FUSED");
  if (e && atoi(e) == 0) return false;
  if (progs.size() < 2 || progs.size() > VX_JIT_FUSED_MAX) return false;
  const JitFusedPlan P = jit_fused_plan(progs, nch);
  return P.max_col >= 1 && (size_t)P.max_col * 512 + (size_t)P.waves * 1024 <= 80 * 1024;
}
static std::string jit_fused_source(const std::vector<const uint64_t*>& progs, int nch) {
  const JitFusedPlan P = jit_fused_plan(progs, nch);
  const int W = P.waves;
  const size_t lds = (size_t)P.max_col * 512 + (size_t)W * 1024;
  const int blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>(160 * 1024 / lds, 2));
  const int waves_per_simd = std::max(1, std::min(8, blocks_per_cu * W / 4));
  std::ostringstream s;
  s << "typedef unsigned long long uint64_t;\ntypedef unsigned int uint32_t;\n"
    << "#define VX_ALPHA_POWS " << VX_ALPHA_POWS << "\n#define VX_JIT_FUSED_MAX " << VX_JIT_FUSED_MAX << "\n#define FW " << W << "\n#define FCOLS " << P.max_col
    << "\n#define FLOADS " << (P.max_col + W - 1) / W << "\n#define FWPS " << waves_per_simd << "\n"
    << jit_limits_defines() << JIT_PRELUDE << JIT_DOT3 << R"VXJIT(
struct JitGateRt {
  int gate_index, selector_index, group_start, group_end;
};
struct JitFusedParams {
  const u64 *cs, *wires;
  const Limbs3x2* alpha_limbs;
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch;
  int base_idx, ngates;
  int const_base, pad_;
  u64 pih[4];
  u64 zh_inv[VX_MAX_RATE];
  JitGateRt g[VX_JIT_FUSED_MAX];
};
extern "C" __global__ __launch_bounds__(64 * FW, FWPS) void vx_program_gates_fused(JitFusedParams p) {
  __shared__ u64 LW[FCOLS * 64];
  __shared__ u64 LT[FW * 2 * 64];
  const u32 lane0 = threadIdx.x & 63u, lane = lane0;
  const u32 wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t il_raw = (size_t)blockIdx.x * 64 + lane;
  const bool live = il_raw < p.rows;
  const size_t il = live ? il_raw : p.rows - 1;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const u64* __restrict__ CS = p.cs;
  const u64* __restrict__ W = p.wires;
  const Limbs3x2* AL0 = p.alpha_limbs + p.base_idx;
  const int nsel = p.num_selectors, cbase = p.const_base;
  {
    u64 v[FLOADS];
#pragma unroll
    for (int k = 0; k < FLOADS; ++k) {
      const u32 col = wid + (u32)k * FW;
      v[k] = col < FCOLS ? W[(size_t)col * SW + il] : 0;
    }
#pragma unroll
    for (int k = 0; k < FLOADS; ++k) {
      const u32 col = wid + (u32)k * FW;
      if (col < FCOLS) LW[col * 64 + lane] = gl_canon(v[k]);
    }
  }
  __syncthreads();
  u64 t0 = 0, t1 = 0;
)VXJIT";
  for (int w = 0; w < W; ++w) {
    s << (w ? "  else if" : "  if") << " (wid == " << w << "u) {\n";
    const std::vector<size_t>& mine = P.of_wave[w];
    for (size_t at = 0; at < mine.size();) {
      const size_t q = P.items[mine[at]].gate;
      s << "  {  // program gate, slot " << q << "\n"
           "    const JitGateRt G = p.g[" << q << "];\n"
           "    const u64 s = CS[(size_t)G.selector_index * N + i];\n"
           "    u64 filter = 1;\n"
           "    for (int q = G.group_start; q < G.group_end; ++q)\n"
           "      if (q != G.gate_index) filter = gl_mul(filter, gl_sub((u64)q, s));\n"
           "    if (nsel > 1) filter = gl_mul(filter, gl_sub(0xFFFFFFFFULL, s));\n"
           "    dot3 A0 = {0, 0, 0}, A1 = {0, 0, 0};\n";
      for (; at < mine.size() && P.items[mine[at]].gate == q; ++at) {
        const JitFusedItem& it = P.items[mine[at]];
        s << "    {  // constraints from " << it.push_begin << " on (estimated " << it.cost << " instructions)\n"
             "    u64 R[VX_PROGRAM_REGS];\n"
             // the same alpha-power limbs / the same wire cells are read by the blocks of different waves: laundering the two bases per
             // block keeps the compiler from hoisting those loads above the wave branch (it did: 248 SGPR and 285 VGPR spills)
             "    const Limbs3x2* AL_ = AL0; asm volatile(\"\" : \"+s\"(AL_)); const Limbs3x2* __restrict__ AL = AL_;\n"
             "    u32 lane = lane0; asm volatile(\"\" : \"+v\"(lane));\n";
        g_jit_lds_wires = true;
        jit_emit_code(s, it.code, nch, false, 0, it.push_begin);
        g_jit_lds_wires = false;
        s << "    }\n";
      }
      s << "    t0 = gl_mad(filter, dot3_reduce_nc(A0), t0);\n";
      if (nch > 1) s << "    t1 = gl_mad(filter, dot3_reduce_nc(A1), t1);\n";
      s << "    (void)A1;\n  }\n";
    }
    s << "  }\n";
  }
  s << R"VXJIT(
  LT[(wid * 2 + 0) * 64 + lane] = t0;
  LT[(wid * 2 + 1) * 64 + lane] = t1;
  __syncthreads();
  if (wid == 0u && live) {
    const u32 z = (u32)(i >> p.log_n);
    const u32 r = p.rate_bits ? (__brev(z) >> (32 - p.rate_bits)) : 0u;
    const u64 zi = p.zh_inv[r];
    u64 a0 = LT[lane];
#pragma unroll
    for (int w = 1; w < FW; ++w) a0 = gl_add(a0, LT[(w * 2 + 0) * 64 + lane]);
    { u64* o = p.out + il; *o = gl_add(*o, gl_mul(a0, zi)); }
)VXJIT";
  if (nch > 1)
    s << "    u64 a1 = LT[64 + lane];\n#pragma unroll\n    for (int w = 1; w < FW; ++w) a1 = gl_add(a1, LT[(w * 2 + 1) * 64 + lane]);\n"
         "    { u64* o = p.out + SW + il; *o = gl_add(*o, gl_mul(a1, zi)); }\n";
  s << "  }\n  (void)cbase; (void)W;\n}\n";
  return s.str();
}

// ---- AIR programs are compiled in CHUNKS (round 3) --------------------------------------------------------------------------
// One kernel for a chip-sized AIR (16 k instructions, 2 k constraints: vectorx_amd/sha256_air.py) took hiprtc SEVEN MINUTES —
// the back end's scheduling and register allocation are superlinear in the size of a basic block.  The program is cut at
// constraint boundaries into chunks of <= VX_JIT_AIR_CHUNK instructions, one kernel each:
//   * a chunk starts with the PROLOGUE that re-creates the registers it reads but does not write — the backward slice of the
//     earlier code (row-type selectors, constants, a running sum ...), replayed in program order;
//   * the Horner accumulation  acc = acc alpha + c_i  is local to the chunk and its result enters the total multiplied by
//     alpha^(K - b) (K constraints in all, b = the first constraint AFTER the chunk) — the host supplies that power — so the
//     kernels simply ADD into the quotient buffer, in any order;
//   * the per-row factors of the constraint kinds (z_last, L_first, L_last: one field inversion per row) are computed once
//     by air_row_factors_kernel (stark.hip.h) and read by every chunk.
#ifndef VX_JIT_AIR_CHUNK
#define VX_JIT_AIR_CHUNK 1200
#endif
struct JitAirChunk {
  std::string src;
  int push_begin = 0, push_end = 0;   // constraints [push_begin, push_end) of the program
  int prologue = 0, body = 0;         // instruction counts (diagnostics)
};
static std::string jit_air_chunk_source(const std::vector<JitIns>& code, int nch, int ncols) {
  std::ostringstream s;
  s << "typedef unsigned long long uint64_t;\ntypedef unsigned int uint32_t;\n#define VX_JIT_BLOCKS_PER_CU 4\n"
    << jit_limits_defines() << JIT_PRELUDE << R"VXJIT(
struct AirParams {
  const u64* trace;
  const u64* aux;
  const u64* program;
  size_t stride;
  size_t rows;
  size_t row_base;
  int log_n, rate_bits, qbits, ncols, nch, npi;
  const u64 *root_lo, *root_hi;
  u64 alphas[VX_MAX_CHALLENGES];
  u64 pi[VX_AIR_MAX_PI];
  u64 chal[VX_AIR_MAX_CHALLENGES];
  u64 zh[VX_MAX_RATE], zh_inv[VX_MAX_RATE];
  u64 last, n_inv;
  u64* out;
  u64* rowfac;
  u64 tail_pow[VX_MAX_CHALLENGES];
};
extern "C" __global__ __launch_bounds__(256, VX_JIT_BLOCKS_PER_CU) void vx_air_quotient(AirParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t n = (size_t)1 << p.log_n;
  const u32 z = (u32)(il >> p.log_n);
  const u32 r = (u32)(il & (n - 1));
  const u32 rn = __brev(((__brev(r) >> (32 - p.log_n)) + 1) & (u32)(n - 1)) >> (32 - p.log_n);
  const size_t il_next = ((size_t)z << p.log_n) | rn;
  const u64 z_last = p.rowfac[il], l_first = p.rowfac[p.rows + il], l_last = p.rowfac[2 * p.rows + il];
  const u64* __restrict__ W = p.trace;
  const u64* __restrict__ AUX = p.aux;
  const size_t SW = p.stride;
  u64 a0 = 0, a1 = 0;
  u64 R[VX_PROGRAM_REGS];
)VXJIT";
  jit_emit_code(s, code, nch, true, ncols);
  s << "  (void)AUX; (void)z_last; (void)l_first; (void)l_last;\n"
       "  { u64* o = p.out + il; *o = gl_add(*o, gl_mul(gl_mul_nc(a0, p.tail_pow[0]), p.zh_inv[z])); }\n";
  if (nch > 1) s << "  { u64* o = p.out + p.rows + il; *o = gl_add(*o, gl_mul(gl_mul_nc(a1, p.tail_pow[1]), p.zh_inv[z])); }\n";
  s << "}\n";
  return s.str();
}
static std::vector<JitAirChunk> jit_air_chunks(const uint64_t* prog, int nch, int ncols) {
  const std::vector<JitIns> code = jit_decode(prog);
  const char* env = getenv("VX_JIT_AIR_CHUNK");
  const size_t limit = env && atoi(env) > 0 ? (size_t)atoi(env) : (size_t)VX_JIT_AIR_CHUNK;
  auto srcs = [](const JitIns& I, int out[2]) {
    int k = 0;
    if (I.op == VX_OP_ADD || I.op == VX_OP_SUB || I.op == VX_OP_MUL) out[k++] = I.a, out[k++] = I.b;
    else if (I.op == VX_OP_PUSH) out[k++] = I.a;
    return k;
  };
  std::vector<JitAirChunk> chunks;
  size_t begin = 0;
  int pushes = 0;
  while (begin < code.size()) {
    // the chunk ends after the first PUSH at or beyond `limit` instructions (or at the end of the program)
    size_t end = begin;
    int np = 0;
    while (end < code.size()) {
      const bool push = code[end].op == VX_OP_PUSH;
      ++end;
      if (push) {
        ++np;
        if (end - begin >= limit) break;
      }
    }
    if (end < code.size()) {  // never leave a tail without constraints for a chunk of its own
      bool more = false;
      for (size_t i = end; i < code.size(); ++i) more = more || code[i].op == VX_OP_PUSH;
      if (!more) end = code.size();
    }
    // registers the body reads before writing them
    bool written[VX_PROGRAM_REGS] = {false}, need[VX_PROGRAM_REGS] = {false};
    for (size_t i = begin; i < end; ++i) {
      int sr[2];
      const int k = srcs(code[i], sr);
      for (int q = 0; q < k; ++q)
        if (!written[sr[q]]) need[sr[q]] = true;
      if (code[i].op != VX_OP_PUSH) written[code[i].dst] = true;
    }
    // backward slice over the earlier code: the nearest earlier definition of every needed register, recursively
    std::vector<char> take(begin, 0);
    for (size_t i = begin; i-- > 0;) {
      const JitIns& I = code[i];
      if (I.op == VX_OP_PUSH || !need[I.dst]) continue;
      take[i] = 1;
      need[I.dst] = false;
      int sr[2];
      const int k = srcs(I, sr);
      for (int q = 0; q < k; ++q) need[sr[q]] = true;
    }
    std::vector<JitIns> sub;
    for (size_t i = 0; i < begin; ++i)
      if (take[i]) sub.push_back(code[i]);
    JitAirChunk c;
    c.prologue = (int)sub.size();
    sub.insert(sub.end(), code.begin() + begin, code.begin() + end);
    c.body = (int)(end - begin);
    c.push_begin = pushes;
    c.push_end = pushes + np;
    c.src = jit_air_chunk_source(sub, nch, ncols);
    chunks.push_back(std::move(c));
    pushes += np;
    begin = end;
  }
  return chunks;
}

struct JitCache {
  std::mutex mu;
  std::map<std::string, std::vector<char>> code;                     // source -> code object
  std::map<std::pair<std::string, int>, hipFunction_t> functions;    // (source, device) -> loaded kernel
};
static JitCache& jit_cache() {
  static JitCache c;
  return c;
}

// hiprtc-compile one source to a gfx950 code object; empty vector + *why on failure.
static std::vector<char> jit_compile(JitApi& api, const std::string& src, std::string* why) {
  hiprtcProgram pr;
  if (api.create(&pr, src.c_str(), "vx_program_gates.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *why = "hiprtcCreateProgram failed";
    return {};
  }
  const char* ol = getenv("VX_JIT_OPT");   // experiments only: the optimisation level handed to hiprtc
  const char* opts[] = {"--offload-arch=gfx950", ol ? ol : "-O3", "-std=c++17"};
  hiprtcResult rc = api.compile(pr, 3, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    api.log_size(pr, &ls);
    std::string log(ls, 0);
    if (ls) api.log(pr, &log[0]);
    *why = "hiprtc compile failed: " + log.substr(0, 400);
    api.destroy(&pr);
    return {};
  }
  size_t cs = 0;
  api.code_size(pr, &cs);
  std::vector<char> code(cs);
  api.code(pr, code.data());
  api.destroy(&pr);
  return code;
}
// The on-disk cache is only used when the directory is the caller's own and closed to others (mode 0700, owner = euid):
// a code object read from it is executed on the GPU, so a directory other users can write to would let them inject code.
static std::string jit_cache_file(JitApi& api, const char* dir, const std::string& src) {
  // key: FNV-1a over the source (which embeds the prelude, so a library update misses) + the hiprtc version and the
  // offload arch the blob was built with/for, so a ROCm upgrade never reuses a stale code object
  uint64_t h = 1469598103934665603ULL;
  for (unsigned char ch : src) h = (h ^ ch) * 1099511628211ULL;
  int rtc_major = 0, rtc_minor = 0;
  if (api.version) api.version(&rtc_major, &rtc_minor);
  char name[128];
  snprintf(name, sizeof name, "/vxjit-gfx950-rtc%d.%d-%016llx-%zu.hsaco", rtc_major, rtc_minor, (unsigned long long)h, src.size());
  return std::string(dir) + name;
}
static bool jit_cache_dir_ok(const char* dir) {
  struct stat st;
  if (stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  return st.st_uid == geteuid() && (st.st_mode & 0077) == 0;
}

// Returns the kernel for this gate set on `device`, or nullptr (with *why set) when it cannot be compiled / loaded.
static hipFunction_t jit_get_kernel(const std::string& src, const char* kernel_name, int device, std::string* why);
static hipFunction_t jit_get_gates(const std::vector<const uint64_t*>& progs, int nch, int device, std::string* why) {
  return jit_get_kernel(jit_gate_source(progs, nch), "vx_program_gate", device, why);
}
static hipFunction_t jit_get_fused_gates(const std::vector<const uint64_t*>& progs, int nch, int device, std::string* why) {
  return jit_get_kernel(jit_fused_source(progs, nch), "vx_program_gates_fused", device, why);
}
// compile (or find in the caches) every chunk WITHOUT loading it: needs no GPU — the `build` step of a host that proves later
static int jit_air_precompile(const uint64_t* prog, int nch, int ncols, int* nchunks, std::string* why);
// the compiled evaluator of one AIR program (stark.hip.h): one kernel per chunk; empty -> interpreter
struct JitAirKernel {
  hipFunction_t fn;
  int push_begin, push_end;
};
static std::vector<JitAirKernel> jit_air_get(const uint64_t* prog, int nch, int ncols, int device, std::string* why) {
  std::vector<JitAirKernel> out;
  if (getenv("VX_NO_JIT")) {
    *why = "VX_NO_JIT is set";
    return out;
  }
  if (!jit_api().ok) {            // before a megabyte of chunk sources is generated for nothing
    *why = "libhiprtc.so not available";
    return out;
  }
  // per (program, nch, ncols, device): the loaded kernels — generating a megabyte of source per proof just to look it up again
  // cost more than the kernels take to run.  One entry per key, compiled ONCE under the entry's own lock (the map's lock is held for
  // the lookup only: another lane proving another table is not blocked for the minutes a compilation can take), and a FAILED
  // compilation / load is remembered too — the program then stays on the interpreter without retrying on every proof.
  struct Entry {
    std::mutex m;
    bool done = false;
    std::vector<JitAirKernel> kernels;
    std::string why;
  };
  static std::mutex mu;
  static std::map<std::vector<uint64_t>, std::shared_ptr<Entry>> loaded;
  std::vector<uint64_t> key = {(uint64_t)nch, (uint64_t)ncols, (uint64_t)device};
  for (int pc = 0;; ++pc) {
    key.push_back(prog[pc]);
    if ((prog[pc] & 0xFF) == VX_OP_END) break;
    if ((prog[pc] & 0xFF) == VX_OP_LDI) key.push_back(prog[++pc]);
  }
  std::shared_ptr<Entry> e;
  {
    std::lock_guard<std::mutex> lk(mu);
    std::shared_ptr<Entry>& slot = loaded[key];
    if (!slot) slot = std::make_shared<Entry>();
    e = slot;
  }
  std::lock_guard<std::mutex> lk(e->m);
  if (e->done) {
    if (e->kernels.empty()) *why = e->why;
    return e->kernels;
  }
  e->done = true;
  if (jit_air_precompile(prog, nch, ncols, nullptr, why) < 0) {   // all missing chunks
    e->why = *why;
    return out;
  }
  for (const JitAirChunk& c : jit_air_chunks(prog, nch, ncols)) {
    hipFunction_t fn = jit_get_kernel(c.src, "vx_air_quotient", device, why);
    if (!fn) {
      e->why = *why;
      return {};
    }
    out.push_back(JitAirKernel{fn, c.push_begin, c.push_end});
  }
  e->kernels = out;
  return out;
}
static hipFunction_t jit_get_kernel(const std::string& src, const char* kernel_name, int device, std::string* why) {
  if (getenv("VX_NO_JIT")) {
    *why = "VX_NO_JIT is set";
    return nullptr;
  }
  JitApi& api = jit_api();
  if (!api.ok) {
    *why = "libhiprtc.so not available";
    return nullptr;
  }
  JitCache& C = jit_cache();
  std::lock_guard<std::mutex> lk(C.mu);
  auto fit = C.functions.find({src, device});
  if (fit != C.functions.end()) return fit->second;
  auto cit = C.code.find(src);
  // optional on-disk cache of code objects (VX_JIT_CACHE_DIR): a host that restarts does not recompile its circuits
  std::string cache_file;
  bool from_disk = false;
  if (const char* dir = getenv("VX_JIT_CACHE_DIR")) {
    if (jit_cache_dir_ok(dir)) {
      cache_file = jit_cache_file(api, dir, src);
      if (cit == C.code.end()) {
        if (FILE* f = fopen(cache_file.c_str(), "rb")) {
          std::vector<char> code;
          char buf[65536];
          size_t n;
          while ((n = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + n);
          fclose(f);
          if (code.size() > 64) {
            cit = C.code.emplace(src, std::move(code)).first;
            from_disk = true;
          }
        }
      }
    }
  }
  auto write_cache = [&](const std::vector<char>& code) {
    if (cache_file.empty()) return;  // write-then-rename so that a concurrent reader never sees a partial file
    const std::string tmp = cache_file + ".tmp" + std::to_string((long)getpid());
    if (FILE* f = fopen(tmp.c_str(), "wb")) {
      const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
      fclose(f);
      if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) remove(tmp.c_str());
    }
  };
  if (cit == C.code.end()) {
    std::vector<char> code = jit_compile(api, src, why);
    if (code.empty()) return nullptr;
    write_cache(code);
    cit = C.code.emplace(src, std::move(code)).first;
  }
  hipModule_t mod;
  if (hipModuleLoadData(&mod, cit->second.data()) != hipSuccess) {
    (void)hipGetLastError();
    if (!from_disk) {
      *why = "hipModuleLoadData failed";
      return nullptr;
    }
    // a stale or corrupt blob from the disk cache: drop it (file and memory), compile afresh, try once more — otherwise
    // this gate set would stay on the ~60x slower interpreter in every process that shares the cache
    remove(cache_file.c_str());
    C.code.erase(cit);
    std::vector<char> code = jit_compile(api, src, why);
    if (code.empty()) return nullptr;
    write_cache(code);
    cit = C.code.emplace(src, std::move(code)).first;
    if (hipModuleLoadData(&mod, cit->second.data()) != hipSuccess) {
      (void)hipGetLastError();
      *why = "hipModuleLoadData failed";
      return nullptr;
    }
  }
  hipFunction_t fn;
  if (hipModuleGetFunction(&fn, mod, kernel_name) != hipSuccess) {
    (void)hipGetLastError();
    *why = "hipModuleGetFunction failed";
    return nullptr;
  }
  C.functions[{src, device}] = fn;
  return fn;
}

// compile every source that is in neither cache (process map, VX_JIT_CACHE_DIR): no device needed.  Returns how many were compiled.
static int jit_precompile_sources(const std::vector<const std::string*>& srcs, std::string* why) {
  JitApi& api = jit_api();
  if (!api.ok) {
    *why = "libhiprtc.so not available";
    return -1;
  }
  JitCache& C = jit_cache();
  const char* dir = getenv("VX_JIT_CACHE_DIR");
  const bool disk = dir && jit_cache_dir_ok(dir);
  std::vector<size_t> todo;
  for (size_t i = 0; i < srcs.size(); ++i) {
    std::lock_guard<std::mutex> lk(C.mu);
    if (C.code.count(*srcs[i])) continue;
    if (disk) {
      if (FILE* f = fopen(jit_cache_file(api, dir, *srcs[i]).c_str(), "rb")) {
        std::vector<char> code;
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + n);
        fclose(f);
        if (code.size() > 64) {
          C.code.emplace(*srcs[i], std::move(code));
          continue;
        }
      }
    }
    todo.push_back(i);
  }
  // independent translation units: a few host threads (VX_JIT_THREADS, default 1: hiprtc 7.x serialises compilations behind
  // one lock — measured 48 s with 1, 4 and 8 threads for the 13 chunks of the SHA-256 AIR)
  const char* te = getenv("VX_JIT_THREADS");
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>(todo.size(), te && atoi(te) > 0 ? (size_t)atoi(te) : 1));
  std::vector<std::vector<char>> objs(todo.size());
  std::vector<std::string> errs(todo.size());
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= todo.size()) return;
      objs[k] = jit_compile(api, *srcs[todo[k]], &errs[k]);
    }
  };
  std::vector<std::thread> pool;
  for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  for (size_t k = 0; k < todo.size(); ++k) {
    if (objs[k].empty()) {
      *why = errs[k];
      return -1;
    }
    if (disk) {
      const std::string cache_file = jit_cache_file(api, dir, *srcs[todo[k]]), tmp = cache_file + ".tmp" + std::to_string((long)getpid());
      if (FILE* f = fopen(tmp.c_str(), "wb")) {
        const bool ok = fwrite(objs[k].data(), 1, objs[k].size(), f) == objs[k].size();
        fclose(f);
        if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) remove(tmp.c_str());
      }
    }
    std::lock_guard<std::mutex> lk(C.mu);
    C.code.emplace(*srcs[todo[k]], std::move(objs[k]));
  }
  return (int)todo.size();
}
static int jit_air_precompile(const uint64_t* prog, int nch, int ncols, int* nchunks, std::string* why) {
  if (!jit_api().ok && !nchunks) {   // nothing could be compiled: do not generate the sources (vx_stark_precompile still reports the chunk count)
    *why = "libhiprtc.so not available";
    return -1;
  }
  const std::vector<JitAirChunk> chunks = jit_air_chunks(prog, nch, ncols);
  if (nchunks) *nchunks = (int)chunks.size();
  std::vector<const std::string*> srcs;
  for (const JitAirChunk& c : chunks) srcs.push_back(&c.src);
  return jit_precompile_sources(srcs, why);
}
// the gate programs of a circuit description (vx_circuit_precompile): one source per program gate
static int jit_gates_precompile(const std::vector<const uint64_t*>& progs, int nch, std::string* why) {
  std::vector<std::string> keep;
  if (jit_fused_applicable(progs, nch)) {   // what vx_circuit_create will ask for
    keep.push_back(jit_fused_source(progs, nch));
  } else {
    for (const std::vector<size_t>& grp : jit_gate_groups(progs)) {
      std::vector<const uint64_t*> sub;
      for (size_t q : grp) sub.push_back(progs[q]);
      keep.push_back(jit_gate_source(sub, nch));
    }
  }
  std::vector<const std::string*> srcs;
  for (const std::string& k : keep) srcs.push_back(&k);
  return jit_precompile_sources(srcs, why);
}

// ---- the second-round column programs (aux.hip.h / vx_stark_aux_columns) compiled like AIR programs: one kernel per program ----
static std::string jit_aux_source(const uint64_t* prog, int nfrac) {
  std::ostringstream s;
  s << "typedef unsigned long long uint64_t;\ntypedef unsigned int uint32_t;\n" << jit_limits_defines() << "#define VX_AUX_MAX_CHALLENGES " << VX_AUX_MAX_CHALLENGES
    << "\n" << JIT_PRELUDE << R"VXJIT(
struct AuxFracParams {
  const u64* trace;
  const u64* program;
  const int* frac_out;
  u64* out;
  size_t n;
  int ncols, nfrac;
  int parts;
  u64 chal[VX_AUX_MAX_CHALLENGES];
};
__device__ __noinline__ u64 aux_inv(u64 a) { return gl_inv(a); }
extern "C" __global__ __launch_bounds__(256) void vx_aux_fractions(AuxFracParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.n) return;
  const u64* __restrict__ W = p.trace;
  const size_t SW = p.n;
  u64 R[VX_PROGRAM_REGS];
)VXJIT";
  g_jit_aux.on = true;
  g_jit_aux.nfrac = nfrac;
  jit_emit_code(s, jit_decode(prog), 1, false, 0);
  g_jit_aux.on = false;
  s << "}\n";
  return s.str();
}
static int jit_aux_precompile(const uint64_t* prog, int nfrac, std::string* why) {
  if (!jit_api().ok) {
    *why = "libhiprtc.so not available";
    return -1;
  }
  const std::string src = jit_aux_source(prog, nfrac);
  return jit_precompile_sources({&src}, why);
}
// the loaded kernel of an aux program on `device`, or nullptr (interpreter): compiled once per (program, device), failures remembered
static hipFunction_t jit_aux_get(const uint64_t* prog, int nfrac, int device, std::string* why) {
  if (getenv("VX_NO_JIT")) {
    *why = "VX_NO_JIT is set";
    return nullptr;
  }
  if (!jit_api().ok) {
    *why = "libhiprtc.so not available";
    return nullptr;
  }
  struct Entry {
    std::mutex m;
    bool done = false;
    hipFunction_t fn = nullptr;
    std::string why;
  };
  static std::mutex mu;
  static std::map<std::vector<uint64_t>, std::shared_ptr<Entry>> loaded;
  std::vector<uint64_t> key = {(uint64_t)nfrac, (uint64_t)device};
  for (int pc = 0;; ++pc) {
    key.push_back(prog[pc]);
    if ((prog[pc] & 0xFF) == VX_OP_END) break;
    if ((prog[pc] & 0xFF) == VX_OP_LDI) key.push_back(prog[++pc]);
  }
  std::shared_ptr<Entry> e;
  {
    std::lock_guard<std::mutex> lk(mu);
    std::shared_ptr<Entry>& slot = loaded[key];
    if (!slot) slot = std::make_shared<Entry>();
    e = slot;
  }
  std::lock_guard<std::mutex> lk(e->m);
  if (e->done) {
    if (!e->fn) *why = e->why;
    return e->fn;
  }
  e->done = true;
  e->fn = jit_get_kernel(jit_aux_source(prog, nfrac), "vx_aux_fractions", device, why);
  if (!e->fn) e->why = *why;
  return e->fn;
}

// Constraint programs compiled to native gfx950 code when a circuit is built.
//
// A gate outside the native set arrives as a straight-line constraint program (include/vxprover.h VX_OP_*).  The
// interpreter (program_gates_kernel) keeps its 64 virtual registers in per-thread scratch memory and pays ~400 cycles
// per interpreted instruction per wavefront; a recursion circuit carries thousands of program words, which would cost
// more than the whole native quotient kernel.  So vx_circuit_create turns every program into HIP source — the same
// field primitives (goldilocks.hip.h, embedded at build time as jit_prelude.inc), one line per instruction, virtual
// registers as a local array with constant indices that the compiler promotes to VGPRs — compiles it with hiprtc for
// gfx950 and loads it as a code-object module; prove() then launches one such kernel per program gate.  Everything
// that differs between circuits using the same gate (selector column, group range, position of the gate's constraints
// in the alpha-power table, ...) is a RUNTIME argument, so a program is compiled once per process, whatever circuit it
// appears in.  hiprtc is reached through dlopen: when it is missing, or VX_NO_JIT=1, or a compile fails, the gate
// stays on the interpreter — still on the GPU, same values (tests/test_gpu_prover.py compares the two paths).
#pragma once
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

struct JitGateParams {  // mirrored textually in jit_source_header()
  const u64 *cs, *wires;
  const u64* alpha_pows;
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch;
  int gate_index, selector_index, group_start, group_end, base_idx, pad;
  u64 pih[4];
  u64 zh_inv[VX_MAX_RATE];
};

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
#undef VX_JIT_SYM
    api.ok = api.create && api.compile && api.log_size && api.log && api.code_size && api.code && api.destroy;
  });
  return api;
}

// HIP source of one program (the words up to and including VX_OP_END; already validated by circuit_create).
static std::string jit_source(const uint64_t* prog, int nch) {
  std::ostringstream s;
  s << "typedef unsigned long long uint64_t;\ntypedef unsigned int uint32_t;\n" << JIT_PRELUDE << R"VXJIT(
struct JitGateParams {
  const u64 *cs, *wires;
  const u64* alpha_pows;
  u64* out;
  size_t N, rows, row_base, stride_w;
  int log_n, rate_bits, num_selectors, nch;
  int gate_index, selector_index, group_start, group_end, base_idx, pad;
  u64 pih[4];
  u64 zh_inv[16];
};
extern "C" __global__ __launch_bounds__(256) void vx_program_gate(JitGateParams p) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.rows) return;
  const size_t N = p.N, SW = p.stride_w, i = il + p.row_base;
  const u32 z = (u32)(i >> p.log_n);
  const u32 r = p.rate_bits ? (__brev(z) >> (32 - p.rate_bits)) : 0u;
  const u64* __restrict__ CS = p.cs;
  const u64* __restrict__ W = p.wires;
  const u64* __restrict__ AP = p.alpha_pows + p.base_idx;
  const u64 s = CS[(size_t)p.selector_index * N + i];
  u64 filter = 1;
  for (int q = p.group_start; q < p.group_end; ++q)
    if (q != p.gate_index) filter = gl_mul(filter, gl_sub((u64)q, s));
  if (p.num_selectors > 1) filter = gl_mul(filter, gl_sub(0xFFFFFFFFULL, s));
  const int nsel = p.num_selectors;
  u64 a0 = 0, a1 = 0;
  u64 R[64];
)VXJIT";
  int k = 0;
  for (int pc = 0;; ++pc) {
    const uint64_t ins = prog[pc];
    const int op = (int)(ins & 0xFF), dst = (int)((ins >> 8) & 63), a = (int)((ins >> 16) & 0xFFFF), b = (int)((ins >> 32) & 0xFFFF);
    if (op == VX_OP_END) break;
    switch (op) {
      case VX_OP_LDW: s << "  R[" << dst << "] = gl_canon(W[(size_t)" << a << " * SW + il]);\n"; break;
      case VX_OP_LDC: s << "  R[" << dst << "] = CS[(size_t)(nsel + " << a << ") * N + i];\n"; break;
      case VX_OP_LDI: s << "  R[" << dst << "] = " << vxh::canon(prog[++pc]) << "ULL;\n"; break;
      case VX_OP_ADD: s << "  R[" << dst << "] = gl_add(R[" << (a & 63) << "], R[" << (b & 63) << "]);\n"; break;
      case VX_OP_SUB: s << "  R[" << dst << "] = gl_sub(R[" << (a & 63) << "], R[" << (b & 63) << "]);\n"; break;
      case VX_OP_MUL: s << "  R[" << dst << "] = gl_mul(R[" << (a & 63) << "], R[" << (b & 63) << "]);\n"; break;
      case VX_OP_PUSH:
        s << "  a0 = gl_mad(R[" << (a & 63) << "], AP[" << k << "], a0);\n";
        if (nch > 1) s << "  a1 = gl_mad(R[" << (a & 63) << "], AP[" << (VX_ALPHA_POWS + k) << "], a1);\n";
        ++k;
        break;
      case VX_OP_LDP: s << "  R[" << dst << "] = p.pih[" << (a & 3) << "];\n"; break;
      default: break;
    }
  }
  s << "  const u64 zi = p.zh_inv[r];\n"
       "  { u64* o = p.out + il; *o = gl_add(*o, gl_mul(gl_mul(filter, a0), zi)); }\n";
  if (nch > 1) s << "  { u64* o = p.out + SW + il; *o = gl_add(*o, gl_mul(gl_mul(filter, a1), zi)); }\n";
  s << "}\n";
  return s.str();
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

// Returns the kernel for this program on `device`, or nullptr (with *why set) when it cannot be compiled / loaded.
static hipFunction_t jit_get(const uint64_t* prog, int nch, int device, std::string* why) {
  if (getenv("VX_NO_JIT")) {
    *why = "VX_NO_JIT is set";
    return nullptr;
  }
  JitApi& api = jit_api();
  if (!api.ok) {
    *why = "libhiprtc.so not available";
    return nullptr;
  }
  const std::string src = jit_source(prog, nch);
  JitCache& C = jit_cache();
  std::lock_guard<std::mutex> lk(C.mu);
  auto fit = C.functions.find({src, device});
  if (fit != C.functions.end()) return fit->second;
  auto cit = C.code.find(src);
  if (cit == C.code.end()) {
    hiprtcProgram pr;
    if (api.create(&pr, src.c_str(), "vx_program_gate.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      *why = "hiprtcCreateProgram failed";
      return nullptr;
    }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    hiprtcResult rc = api.compile(pr, 3, opts);
    if (rc != HIPRTC_SUCCESS) {
      size_t ls = 0;
      api.log_size(pr, &ls);
      std::string log(ls, 0);
      if (ls) api.log(pr, &log[0]);
      *why = "hiprtc compile failed: " + log.substr(0, 400);
      api.destroy(&pr);
      return nullptr;
    }
    size_t cs = 0;
    api.code_size(pr, &cs);
    std::vector<char> code(cs);
    api.code(pr, code.data());
    api.destroy(&pr);
    cit = C.code.emplace(src, std::move(code)).first;
  }
  hipModule_t mod;
  if (hipModuleLoadData(&mod, cit->second.data()) != hipSuccess) {
    (void)hipGetLastError();
    *why = "hipModuleLoadData failed";
    return nullptr;
  }
  hipFunction_t fn;
  if (hipModuleGetFunction(&fn, mod, "vx_program_gate") != hipSuccess) {
    (void)hipGetLastError();
    *why = "hipModuleGetFunction failed";
    return nullptr;
  }
  C.functions[{src, device}] = fn;
  return fn;
}

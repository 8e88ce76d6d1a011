/* Plain-C consumer of the drop-in boundary (include/vxprover.h): what a cgo / Rust-FFI / C++ caller does, with no
 * Python in the loop.  Builds a synthetic circuit with libvxsynth, loads it (vx_circuit_create), proves twice
 * (host witness, then device-resident witness), checks the two proofs are identical and prints an FNV-1a hash of the
 * proof bytes so the pytest wrapper can compare it with the proof obtained through the ctypes mirror.
 *
 *   gcc -O2 -I include tests/c_abi_smoke.c -o gpurun_out/c_abi_smoke -L vectorx_amd -lvxprover -lvxsynth -Wl,-rpath,$PWD/vectorx_amd
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vxprover.h"

typedef struct vxs_circuit vxs_circuit;
vxs_circuit* vxs_build(int degree_bits, uint64_t seed, int poseidon_percent);
void vxs_free(vxs_circuit*);
const vx_circuit_desc* vxs_desc(vxs_circuit*);
const uint64_t* vxs_witness(vxs_circuit*);

#define CHECK(call)                                                             \
  do {                                                                          \
    int _rc = (call);                                                           \
    if (_rc != VX_OK) {                                                         \
      fprintf(stderr, "%s failed (%d): %s\n", #call, _rc, vx_last_error());     \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

static uint64_t fnv1a(const uint8_t* p, size_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ULL;
  return h;
}

int main(int argc, char** argv) {
  int degree_bits = argc > 1 ? atoi(argv[1]) : 10;
  uint64_t seed = argc > 2 ? strtoull(argv[2], 0, 10) : 7;
  vxs_circuit* sc = vxs_build(degree_bits, seed, 50);
  if (!sc) { fprintf(stderr, "vxs_build failed\n"); return 1; }
  const vx_circuit_desc* desc = vxs_desc(sc);
  const uint64_t* wires = vxs_witness(sc);
  size_t n = (size_t)1 << degree_bits, wbytes = (size_t)desc->num_wires * n * 8;

  vx_ctx* ctx = NULL;
  CHECK(vx_ctx_create(0, &ctx));
  vx_circuit* circuit = NULL;
  CHECK(vx_circuit_create(ctx, desc, &circuit));
  uint64_t digest[4];
  CHECK(vx_circuit_digest(circuit, digest));

  size_t cap = vx_proof_size_bound(circuit), len1 = cap, len2 = cap;
  uint8_t* p1 = malloc(cap);
  uint8_t* p2 = malloc(cap);
  CHECK(vx_prove(ctx, circuit, wires, 0, NULL, p1, &len1));
  void* dw = NULL;
  CHECK(vx_dev_alloc(ctx, wbytes, &dw));
  CHECK(vx_dev_upload(ctx, dw, wires, wbytes));
  CHECK(vx_prove(ctx, circuit, (const uint64_t*)dw, 1, NULL, p2, &len2));
  if (len1 != len2 || memcmp(p1, p2, len1)) { fprintf(stderr, "host- and device-witness proofs differ\n"); return 1; }
  /* circuit.verify(&proof): accepted as produced, rejected with one bit flipped */
  CHECK(vx_verify(circuit, p1, len1));
  p2[len2 / 2] ^= 1;
  if (vx_verify(circuit, p2, len2) != VX_E_PROOF) { fprintf(stderr, "a tampered proof was not rejected\n"); return 1; }
  /* too-small output buffer: required size reported, nothing written past the buffer */
  size_t small = 16;
  uint8_t tiny[16];
  int rc = vx_prove(ctx, circuit, wires, 0, NULL, tiny, &small);
  if (rc != VX_E_INVALID || small != len1) { fprintf(stderr, "short-buffer contract violated (%d, %zu)\n", rc, small); return 1; }
  /* a bad pow hint is an error, not a wrong proof */
  uint64_t bad_hint = 1;
  size_t l3 = cap;
  rc = vx_prove(ctx, circuit, wires, 0, &bad_hint, p2, &l3);
  if (rc == VX_OK) {
    /* 1 happened to be valid (probability 2^-16): then it must at least be a different-or-equal proof */
  } else if (rc != VX_E_INVALID) { fprintf(stderr, "bad hint: unexpected rc %d\n", rc); return 1; }

  printf("OK degree_bits=%d len=%zu fnv=%016llx digest0=%016llx\n", degree_bits, len1, (unsigned long long)fnv1a(p1, len1),
         (unsigned long long)digest[0]);
  CHECK(vx_dev_free(ctx, dw));
  vx_circuit_free(circuit);
  vx_ctx_destroy(ctx);
  vxs_free(sc);
  free(p1);
  free(p2);
  return 0;
}

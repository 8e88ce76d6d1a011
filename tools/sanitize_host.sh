#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over ALL host code the CPU test-suite exercises (no GPU needed, none used):
#   * the oracle (oracle/oracle_capi.cpp — prover + verifier restatement),
#   * the synthetic circuit / witness generator (vectorx_amd/synth/synth_circuit.cpp),
#   * the HOST side of libvxprover.so (description checks, .vxcircuit parser, transcript, the standalone plonk / STARK verifiers,
#     the hiprtc code generator): hipcc instruments the host compile only (-fno-gpu-sanitize; GPU ASan is not available on the pool).
# One sanitizer runtime for all three (clang's, preloaded into python).  The instrumented libraries live under $OUT and are swapped
# in by a small runner; nothing in the tree is modified.
#   tools/sanitize_host.sh [pytest args, default: tests -m "not gpu"]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${VX_SAN_DIR:-/tmp/vx_san}
LLVM=/opt/rocm/lib/llvm
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -shared"
mkdir -p "$OUT"
$LLVM/bin/clang++ $SAN -fopenmp -march=x86-64-v3 -o "$OUT/liboracle_san.so" "$ROOT/oracle/oracle_capi.cpp"
$LLVM/bin/clang++ $SAN -o "$OUT/libvxsynth_san.so" "$ROOT/vectorx_amd/synth/synth_circuit.cpp"
make -C "$ROOT/vectorx_amd/csrc" jit_prelude.inc >/dev/null 2>&1 || true
(cd "$ROOT/vectorx_amd/csrc" && /opt/rocm/bin/hipcc $SAN --offload-arch=gfx950 -fno-gpu-sanitize -w -o "$OUT/libvxprover_san.so" vxprover.hip -ldl)
# round 5: the host preparation and the row writers of the native trace generators (test-only host build, tests/tracegen_host.cpp)
$LLVM/bin/clang++ $SAN -o "$OUT/libtracegen_host_san.so" "$ROOT/tests/tracegen_host.cpp"
cat > "$OUT/run.py" <<EOF
import pathlib, sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import vectorx_amd
vectorx_amd._LIB_PATH = pathlib.Path("$OUT/libvxprover_san.so")
import vectorx_amd.synth as _synth
_synth._SO = pathlib.Path("$OUT/libvxsynth_san.so")
import oracle_lib
oracle_lib.build = lambda: pathlib.Path("$OUT/liboracle_san.so")
import os
os.environ["VX_TRACEGEN_HOST_SO"] = "$OUT/libtracegen_host_san.so"
import pytest
sys.exit(pytest.main(sys.argv[1:]))
EOF
cd "$ROOT"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ $# -eq 0 ]; then set -- tests -m "not gpu"; fi
# -s: a sanitizer report must reach the terminal (pytest's capture would swallow it when the process dies)
LD_PRELOAD=$RT python3 "$OUT/run.py" "$@" -x -q -s -p no:cacheprovider

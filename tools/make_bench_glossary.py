#!/usr/bin/env python3
"""profiles/bench_line_glossary.md from a COMPLETE bench line (bench_line_full.json, written next to every `bench.py` run; `--full-line`
prints it): every description the compact line leaves out, keyed by the path of its field.

    python tools/make_bench_glossary.py gpurun_out/bench_line_full.json [more full lines ...] > profiles/bench_line_glossary.md

bench.py prints the compact form (bench_prove.compact_line: every number, no prose, < 6 KB) because the driver's record keeps the last
8 KB of stdout; what a field MEANS is here.  Several lines may be given (N = 1 and N > 1 runs have different legs): first description wins."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

STATIC = {
    "value": "whole-job proofs/sec, witness resident in HBM when the timed region starts (bench contract); = `value_hbm_resident`",
    "value_end_to_end": "SURVEY §8d's end-to-end figure: witness in page-locked HOST memory -> proof bytes, PCIe inclusive (= value_from_host_witness.value); never `value`",
    "stage_ms_per_step": "HIP-event time per stage per proof on the library's stream.  The stages `quotient_l0_permutation`, `quotient_small_native_gates`, "
                         "`quotient_poseidon_gate`, `quotient_lookup_terms`, `quotient_program_gates_jit` are bracketed INSIDE `quotient_eval` (do not add them to it); "
                         "`qgate_<i>` = program gate i on its own (only with VX_JIT_FUSED=0)",
    "roofline": "the coset-LDE launches (ntt2_pass_kernel: 8 coset NTTs per column, two passes) against HBM peak: achieved = algorithmic bytes (72 n per column) / "
                "HIP-event time; traffic = PMC-measured HBM bytes per algorithmic byte (profiles/pmc_traffic_lde.json) x algorithmic bytes",
    "alu_bound_dominant_kernel": "the kernel that dominates by TIME (Poseidon leaf hashing) against the VALU issue bound: 1024 SIMDs x max clock / best measured "
                                 "mixed-stream cycles per wavefront-instruction (profiles/r04_alu_ceiling.json, r03_ubench_int.md)",
    "cpu_baseline": "the oracle (kind `port`: CPU restatement of plonky2 v0.2.0, OpenMP over `cores` host cores) proving THE BENCH CIRCUIT ITSELF once at the bench size in a "
                    "child process after every GPU leg: `value` = 1 / `seconds`, measured in this run, no scaling (`measured_in_this_run`); `sampled` = the bounded "
                    "sample the line carried until round 5 (2^18 rows, scaled linearly by `scaled_by`) with its stage split",
    "dag_header_range_512": "one whole header_range_512 DAG (64 map 2^18 + 63 reduce 2^16 + 1 outer 2^19 plonky2 proofs) on the worker pool (workers_per_gpu x "
                            "lanes_per_worker), plonky2 proofs only.  Since round 6 the circuits carry the RECURSIVE VERIFIER's gate set in its declared row mix "
                            "(vectorx_amd/synth.py RECURSIVE_VERIFIER_MIX).  dag_seconds = the first pass; dag_seconds_all_passes = [schedule, seconds]...",
    "dag_header_range_512_with_starks": "the same DAG with every job's STARK tables (own AIRs), traces generated on the GPU inside the clock, one synthetic chained request; "
                                        "`output_equals_host_computation`: the outer job's 96 output bytes == hashlib + avail_codec over the same request",
    "recursion_circuits_alone": "the DAG's three circuit sizes proven ALONE (not in the pool) with the recursion-shaped mix: ms per proof, the same size with the two-gate "
                                "stand-in of rounds 1-5, `quotient_by_kernel_ms` = the quotient's kernels (nested in quotient_eval_ms), and at the map size every "
                                "program gate as its own kernel (`map_quotient_by_gate_ms_one_kernel_per_gate`, VX_JIT_FUSED=0: what the fused kernel replaced)",
    "chip_starks": "one lone proof of each chip table through stark_chips.ResidentTable (second-round columns on the GPU in every proof); starky-order transcript "
                   "(`openings_digest` 0); `ms_per_proof_openings_digest` = the same table under this library's tree-hash variant (VX_STARK_OPENINGS_DIGEST)",
    "rotate": "one rotate request end to end on the GPU: plonky2 2^19 + BLAKE2b + two SHA-256 commitment chains + the 300 signatures through the signature bus; `output` = the 32 output bytes",
    "glossary": "this file",
    "dropped_for_size": "how many detail tables were left out of the compact line to stay under 6 KB, in the order of bench_prove.compact_line's `order` "
                        "(unit-test cycle counts, per-stage GB/s, the duplicate host-witness block, ...): they are in bench_line_full.json of the same run",
    "dag_header_range_512/stage_elapsed_ms_per_dag": "HIP-event ELAPSED time per stage, summed over the 128 proofs of ONE MORE pass of the DAG with profiling on every "
                                             "lane (`profiled_pass_seconds`; not one of the timed passes).  Nine lanes share the GPU, so a stage's bracket also contains "
                                             "other lanes' kernels: the sums are lane-seconds of waiting + work (they add up to ~ lanes x wall), NOT kernel time — a "
                                             "kernel that needs much LDS (the fused gate kernel: 77 KB per workgroup) waits longest for a CU; "
                                             "`quotient_elapsed_by_kernel_ms_per_dag` = the quotient's kernels the same way (nested in quotient_eval); kernel times "
                                             "proper: `recursion_circuits_alone`",
}


def main():
    import bench_prove
    seen = {}

    def walk(o, path, key):
        if isinstance(o, dict):
            for k, v in o.items():
                walk(v, path + [k], k)
        elif isinstance(o, list):
            for v in o:
                walk(v, path, key)
        elif isinstance(o, str) and len(o) > 48 and key not in bench_prove._KEEP_TEXT and key not in bench_prove._HEX_KEYS:
            seen.setdefault("/".join(path), o)

    for f in sys.argv[1:]:
        walk(json.loads(Path(f).read_text()), [], None)
    print("# bench.py — what the fields of the line mean\n")
    print("`bench.py` prints ONE compact JSON line (every number, no prose, < 6 KB: the driver's record keeps the last 8 KB of stdout); the complete line of the")
    print("same run — with the descriptions below inside it — is written to `bench_line_full.json` (`--full-line` prints it instead).  Generated by")
    print("`tools/make_bench_glossary.py` from a complete line; hex fields (`root`, `input`, `output`, `proof_sha256`) are cut to 16 characters in the compact line.\n")
    print("## Fields described here only\n")
    for k, v in STATIC.items():
        print(f"* **`{k}`** — {v}")
    print("\n## Descriptions the complete line carries (path of the field -> text)\n")
    for k in sorted(seen):
        print(f"* **`{k}`** — {seen[k]}")


if __name__ == "__main__":
    main()

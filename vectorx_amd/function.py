"""`build` / `prove input.json` — the CLI contract of the reference's function binaries, over the MI355X prover.

The Succinct platform drives the reference with
    ./build/header_range_512 prove input.json        (/root/reference/succinct.json:18; header_range_256: :8, rotate: :27)
after a `build` step that leaves the compiled circuits under ./build (:7, :17, :26).  `Plonky2xFunction::entrypoint()`
(/root/reference/bin/header_range_512.rs:16, bin/rotate.rs:15) parses the request, loads the circuits, proves, and writes
output.json.  This module is that surface with the GPU library underneath:

    python -m vectorx_amd.function build  --function header_range_512 [--build-dir build]
    python -m vectorx_amd.function prove input.json --function header_range_512 [--build-dir build] [--output output.json]

* request  (plonky2x `ProofRequest::Bytes`, recalled — the crate is not vendored):
      {"type": "req_bytes", "releaseId": "...", "data": {"input": "0x<hex>"}}
  header_range input  = abi.encodePacked(uint32 trustedBlock, bytes32 trustedHeader, uint64 authoritySetId,
                        bytes32 authoritySetHash, uint32 targetBlock) = 80 bytes (/root/reference/bin/vectorx.rs:106-112);
  rotate input        = abi.encodePacked(uint64 authoritySetId, bytes32 authoritySetHash) = 40 bytes
                        (/root/reference/circuits/rotate.rs:87-88).
* result   (`ProofResult::Bytes`):  {"type": "res_bytes", "data": {"proof": "0x<hex>", "output": "0x<hex>"}}
  header_range output = abi.encode(bytes32, bytes32, bytes32) = 96 bytes (/root/reference/circuits/header_range.rs:56-58);
  rotate output       = bytes32 (/root/reference/circuits/rotate.rs:108).

What is real here and what is a stand-in.  REAL: the request / result framing and the byte-length checks; the build
artefacts (`<kind>.vxcircuit`, loaded back with vx_circuit_load — nothing is rebuilt at prove time); the shape of the
work — header_range_N = N/8 map + N/8 - 1 reduce + 1 outer plonky2 proofs scheduled layer by layer
(vectorx_amd/mapreduce.py), rotate = one proof; every proof made by vx_prove and checked by vx_verify.  STAND-IN: the
circuits (synthetic gate mix — the real ones need the Rust builder), hence the witness generator and
the meaning of the output bytes: the map jobs' public inputs are derived from the request's input bytes, and the
`output` is derived from the outer proof's digest, so the output is a deterministic function of the input that can
only be produced by proving the whole DAG.

The proving backend is injected (`GpuBackend` by default).  The CPU tests inject their own backend
(tests/test_function_cli.py); this module imports the product library only.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import sys
import time
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import avail_codec as codec
from . import mapreduce as mr

P = mr.P

FUNCTIONS = {
    # name: (input bytes, output bytes, number of map jobs or 0 for a single proof)
    "header_range_256": (80, 96, 32),
    "header_range_512": (80, 96, 64),
    "rotate": (40, 32, 0),
}
CIRCUIT_SEEDS = {"map": 101, "reduce": 202, "outer": 303, "rotate": 404}


@dataclass
class Sizes:
    """log2 rows of the stand-in circuits (the defaults are the DAG benchmark's: tools/dag_bench.py)"""
    map_log_n: int = 18
    reduce_log_n: int = 16
    outer_log_n: int = 19
    rotate_log_n: int = 19
    poseidon_percent: int = 50

    def log_n(self, kind):
        return getattr(self, kind + "_log_n")


class RequestError(ValueError):
    pass


def parse_request(text: str, function: str) -> bytes:
    """input.json -> the raw input bytes, validated against the function's packing."""
    try:
        req = json.loads(text)
    except json.JSONDecodeError as e:
        raise RequestError(f"input.json is not JSON: {e}") from None
    if not isinstance(req, dict) or req.get("type") != "req_bytes":
        raise RequestError('unsupported request type (expected {"type": "req_bytes", ...}; plonky2x `ProofRequest::Bytes`)')
    data = req.get("data")
    if not isinstance(data, dict) or not isinstance(data.get("input"), str):
        raise RequestError('request has no data.input hex string')
    h = data["input"]
    h = h[2:] if h.startswith(("0x", "0X")) else h
    try:
        raw = bytes.fromhex(h)
    except ValueError:
        raise RequestError("data.input is not hex") from None
    want = FUNCTIONS[function][0]
    if len(raw) != want:
        raise RequestError(f"{function}: input is {len(raw)} bytes, the circuit reads {want} "
                           f"({'uint32|bytes32|uint64|bytes32|uint32' if want == 80 else 'uint64|bytes32'}, packed)")
    check_request_semantics(function, raw)
    return raw


def decode_header_range_input(raw: bytes) -> dict:
    """abi.encodePacked(uint32, bytes32, uint64, bytes32, uint32): big-endian integers (bin/vectorx.rs:106-112)."""
    d = codec.unpack_header_range_input(raw)
    return {**d, "trusted_header": d["trusted_header"].hex(), "authority_set_hash": d["authority_set_hash"].hex()}


def check_request_semantics(function: str, raw: bytes) -> None:
    """What the circuit itself would refuse: header_range_N proves the headers (trusted, target] with
    0 < target - trusted <= N (/root/reference/circuits/builder/subchain_verification.rs:29-36: "the range [trusted_block + 1, target_block] inclusive" over at most MAX_NUM_HEADERS headers; the map stage walks
    trusted+1 .. trusted+N and the reduce stage requires the target block among them)."""
    if function.startswith("header_range"):
        d = codec.unpack_header_range_input(raw)
        span, limit = d["target_block"] - d["trusted_block"], int(function.rsplit("_", 1)[1])
        if not 0 < span <= limit:
            raise RequestError(f"{function}: target_block - trusted_block = {span}, the circuit covers 1..{limit} headers")
    else:
        codec.unpack_rotate_input(raw)


def format_result(proof: bytes, output: bytes) -> str:
    return json.dumps({"type": "res_bytes", "data": {"proof": "0x" + proof.hex(), "output": "0x" + output.hex()}})


def kinds_of(function: str):
    return ["rotate"] if FUNCTIONS[function][2] == 0 else ["map", "reduce", "outer"]


# ---- backends ------------------------------------------------------------------------------------------------------
class GpuBackend:
    """vx_circuit_create / vx_circuit_serialize / vx_circuit_load / vx_prove / vx_verify on one MI355X."""

    def __init__(self, device: int = 0):
        import vectorx_amd as vx
        self.vx = vx
        self.ctx = vx.Context(device)      # raises without a GPU: there is no CPU fallback

    def compile(self, desc_ptr):
        """-> (prover file bytes, verifier file bytes)"""
        c = self.vx.Circuit(self.ctx, desc_ptr)
        cap = c.constants_sigmas_cap()
        c.free()
        return (self.vx.circuit_serialize(desc_ptr, cap, with_preprocessed=True),
                self.vx.circuit_serialize(desc_ptr, cap, with_preprocessed=False))

    def load(self, blob: bytes):
        c = self.vx.Circuit.load(self.ctx, blob)
        return c

    def prove(self, circuit, wires: np.ndarray) -> bytes:
        proof = circuit.prove(wires)
        circuit.verify(proof)               # `circuit.verify(&proof, ..)` follows every prove in the reference
        return proof

    def close(self):
        self.ctx.close()


# ---- build ---------------------------------------------------------------------------------------------------------
def build(function: str, build_dir: Path, sizes: Sizes, backend) -> dict:
    """Compile the function's circuits and leave them under build_dir (prover + verifier files + a manifest)."""
    from .synth import SynthCircuit
    build_dir.mkdir(parents=True, exist_ok=True)
    manifest = {"function": function, "sizes": sizes.__dict__, "circuits": {}}
    for kind in kinds_of(function):
        sc = SynthCircuit(sizes.log_n(kind), seed=CIRCUIT_SEEDS[kind], poseidon_percent=sizes.poseidon_percent, witness_seed=0)
        prover_blob, verifier_blob = backend.compile(sc.desc_ptr)
        (build_dir / f"{function}.{kind}.vxcircuit").write_bytes(prover_blob)
        (build_dir / f"{function}.{kind}.verifier.vxcircuit").write_bytes(verifier_blob)
        manifest["circuits"][kind] = {"log_n": sizes.log_n(kind), "bytes": len(prover_blob),
                                      "sha256": hashlib.sha256(prover_blob).hexdigest()}
        sc.free()
    (build_dir / f"{function}.json").write_text(json.dumps(manifest, indent=1))
    return manifest


# ---- prove ---------------------------------------------------------------------------------------------------------
class _FileProver:
    """mapreduce prover over circuits LOADED from the build directory; one synthetic witness per job."""

    def __init__(self, backend, function, kind, build_dir: Path, sizes: Sizes, jobs, seed_material: bytes):
        from .synth import SynthCircuit
        self.backend = backend
        blob = (build_dir / f"{function}.{kind}.vxcircuit").read_bytes()
        self.circuit = backend.load(blob)
        self.kind, self.log_n = kind, sizes.log_n(kind)
        self.sc = {}
        for key in jobs:
            li, j = key
            self.sc[key] = SynthCircuit(self.log_n, seed=CIRCUIT_SEEDS[kind], poseidon_percent=sizes.poseidon_percent,
                                        witness_seed=1000 * li + j + 1)

    def prove(self, key, public_inputs, lane=0):
        sc = self.sc[key]
        r0, r2 = sc.patch_public_inputs(public_inputs)
        w = sc.witness().copy()
        w[:, 0] = r0
        w[:, 2] = r2
        return self.backend.prove(self.circuit, w)

    def free(self):
        for sc in self.sc.values():
            sc.free()
        if hasattr(self.circuit, "free"):
            self.circuit.free()


def prove(function: str, raw_input: bytes, build_dir: Path, backend) -> tuple:
    """-> (proof bytes of the top-level proof, output bytes, stats)"""
    manifest = json.loads((build_dir / f"{function}.json").read_text())
    sizes = Sizes(**manifest["sizes"])
    in_len, out_len, num_map = FUNCTIONS[function]
    assert len(raw_input) == in_len
    seed = hashlib.sha256(function.encode() + raw_input).digest()
    t0 = time.perf_counter()
    rotate_out = None
    if num_map == 0:
        # the data the rotate witness generator fetches (circuits/rotate.rs:95-99) — synthesised: an epoch-end header whose
        # digest carries the next authority set; the host runs the checks the circuit makes and the OUTPUT is the real
        # thing, the chained SHA-256 commitment of the new set (rotate.rs:101-108); the proof's public inputs bind both
        header, start, pubkeys = codec.synthetic_epoch_end_header(seed)
        rotate_out = codec.rotate_output(header, start, pubkeys)
        p = _FileProver(backend, function, "rotate", build_dir, sizes, [(0, 0)], seed)
        proof = p.prove((0, 0), mr.digest_to_field(hashlib.sha256(seed + hashlib.sha256(header).digest() + rotate_out).digest()))
        p.free()
        nproofs = 1
    else:
        spec = mr.DagSpec(num_map=num_map, map_log_n=sizes.map_log_n, reduce_log_n=sizes.reduce_log_n, outer_log_n=sizes.outer_log_n,
                          poseidon_percent=sizes.poseidon_percent)
        provers = []

        def make(kind, log_n, jobs):
            p = _FileProver(backend, function, kind, build_dir, sizes, jobs, seed)
            provers.append(p)
            return p

        res = mr.run_dag(spec, make, input_seed=seed)
        proof = res["my_proofs"][(len(spec.layers()) - 1, 0)]
        nproofs = res["proofs"]
        for p in provers:
            p.free()
    # stand-in output: a deterministic function of the top-level proof (see the module docstring)
    if rotate_out is not None:
        out = rotate_out
    else:
        d = hashlib.sha256(b"vectorx-output" + proof).digest()
        out = b"".join(hashlib.sha256(d + bytes([i])).digest() for i in range(out_len // 32))
    return proof, out, {"proofs": nproofs, "seconds": time.perf_counter() - t0}


def main(argv=None, backend_factory=GpuBackend) -> int:
    ap = argparse.ArgumentParser(prog="vectorx_amd.function", description=__doc__.split("\n\n")[0])
    ap.add_argument("command", choices=["build", "prove"])
    ap.add_argument("input", nargs="?", help="input.json (prove)")
    ap.add_argument("--function", default="header_range_512", choices=sorted(FUNCTIONS))
    ap.add_argument("--build-dir", default="build")
    ap.add_argument("--output", default="output.json")
    for f in ("map_log_n", "reduce_log_n", "outer_log_n", "rotate_log_n"):
        ap.add_argument("--" + f.replace("_", "-"), type=int, default=None)
    args = ap.parse_args(argv)
    build_dir = Path(args.build_dir)
    if args.command == "prove":
        if not args.input:
            print("prove needs the request file (input.json)", file=sys.stderr)
            return 2
        try:
            raw = parse_request(Path(args.input).read_text(), args.function)
        except (RequestError, OSError) as e:
            print(f"error: {e}", file=sys.stderr)
            return 2
        if not (build_dir / f"{args.function}.json").exists():
            print(f"error: {build_dir}/{args.function}.json not found — run `build` first", file=sys.stderr)
            return 2
    backend = backend_factory()
    try:
        if args.command == "build":
            sizes = Sizes()
            for f in ("map_log_n", "reduce_log_n", "outer_log_n", "rotate_log_n"):
                if getattr(args, f) is not None:
                    setattr(sizes, f, getattr(args, f))
            m = build(args.function, build_dir, sizes, backend)
            print(json.dumps(m))
        else:
            proof, out, stats = prove(args.function, raw, build_dir, backend)
            Path(args.output).write_text(format_result(proof, out))
            print(json.dumps({"function": args.function, "output": "0x" + out.hex(), "proof_bytes": len(proof), **stats}))
    finally:
        if hasattr(backend, "close"):
            backend.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

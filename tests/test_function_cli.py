"""The `build` / `prove input.json` -> output.json contract of the reference's function binaries
(/root/reference/succinct.json:7-8,17-18,26-27; input packing /root/reference/bin/vectorx.rs:106-112) over
vectorx_amd/function.py, on the CPU: request / result framing, byte-length checks, artefacts written by `build` and
LOADED by `prove`, the DAG run.  The proving backend here is the oracle (test infrastructure) behind the same
interface the GPU backend implements; tests/test_gpu_boundary.py runs the real `GpuBackend`."""
import ctypes
import json

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd import function as fn


class OracleBackend:
    """CPU stand-in for function.GpuBackend: same four methods, oracle prover (tests only)."""

    def __init__(self):
        self.oracle = oracle_lib.load()
        self.keep = []

    def compile(self, desc_ptr):
        cap = oracle_lib.OracleCircuit(self.oracle, desc_ptr).cap()
        return vx.circuit_serialize(desc_ptr, cap, True), vx.circuit_serialize(desc_ptr, cap, False)

    def load(self, blob):
        pc = vx.ParsedCircuit(blob)
        self.keep.append(pc)
        c = oracle_lib.OracleCircuit(self.oracle, pc.desc_ptr)
        c._pc = pc
        return c

    def prove(self, circuit, wires):
        proof = circuit.prove(wires)
        assert circuit.verify(proof) == ""
        vx.verify_standalone(circuit._pc.desc_ptr, circuit._pc.cap, proof)   # the product verifier, from the file's own data
        return proof


def _request(raw: bytes) -> str:
    return json.dumps({"type": "req_bytes", "releaseId": "test", "data": {"input": "0x" + raw.hex()}})


def _header_range_input(trusted_block=272502, set_id=256, target=272534) -> bytes:
    return (trusted_block.to_bytes(4, "big") + bytes(range(32)) + set_id.to_bytes(8, "big") + bytes(range(32, 64)) + target.to_bytes(4, "big"))


def test_request_parsing_and_packing():
    raw = _header_range_input()
    assert len(raw) == 80
    assert fn.parse_request(_request(raw), "header_range_512") == raw
    d = fn.decode_header_range_input(raw)
    assert d["trusted_block"] == 272502 and d["authority_set_id"] == 256 and d["target_block"] == 272534
    assert fn.parse_request(_request(b"\x07" * 40), "rotate") == b"\x07" * 40
    for bad in ("not json", json.dumps({"type": "req_call"}), json.dumps({"type": "req_bytes", "data": {}}),
                json.dumps({"type": "req_bytes", "data": {"input": "0xzz"}}), _request(raw[:-1]), _request(raw + b"\0")):
        with pytest.raises(fn.RequestError):
            fn.parse_request(bad, "header_range_512")
    with pytest.raises(fn.RequestError):
        fn.parse_request(_request(raw), "rotate")            # 80 bytes is not a rotate input
    out = json.loads(fn.format_result(b"\x01\x02", b"\xaa" * 96))
    assert out == {"type": "res_bytes", "data": {"proof": "0x0102", "output": "0x" + "aa" * 96}}


def test_build_then_prove_rotate(tmp_path):
    b = tmp_path / "build"
    req = tmp_path / "input.json"
    req.write_text(_request(b"\x00" * 7 + b"\x2a" + bytes(range(32))))
    outp = tmp_path / "output.json"
    assert fn.main(["prove", str(req), "--function", "rotate", "--build-dir", str(b), "--output", str(outp)], OracleBackend) == 2  # no build yet
    assert fn.main(["build", "--function", "rotate", "--build-dir", str(b), "--rotate-log-n", "6"], OracleBackend) == 0
    files = sorted(p.name for p in b.iterdir())
    assert files == ["rotate.json", "rotate.rotate.verifier.vxcircuit", "rotate.rotate.vxcircuit"]
    assert fn.main(["prove", str(req), "--function", "rotate", "--build-dir", str(b), "--output", str(outp)], OracleBackend) == 0
    res = json.loads(outp.read_text())
    assert res["type"] == "res_bytes" and len(bytes.fromhex(res["data"]["output"][2:])) == 32
    proof = bytes.fromhex(res["data"]["proof"][2:])
    # the verifier file written by `build` is all a verifier needs
    vc = vx.ParsedCircuit((b / "rotate.rotate.verifier.vxcircuit").read_bytes())
    vx.verify_standalone(vc.desc_ptr, vc.cap, proof)
    # deterministic, and a different input gives a different proof / output
    assert fn.main(["prove", str(req), "--function", "rotate", "--build-dir", str(b), "--output", str(tmp_path / "o2.json")], OracleBackend) == 0
    assert json.loads((tmp_path / "o2.json").read_text()) == res
    req.write_text(_request(b"\x00" * 7 + b"\x2b" + bytes(range(32))))
    assert fn.main(["prove", str(req), "--function", "rotate", "--build-dir", str(b), "--output", str(tmp_path / "o3.json")], OracleBackend) == 0
    assert json.loads((tmp_path / "o3.json").read_text())["data"]["output"] != res["data"]["output"]
    # malformed request: exit code 2, no output written
    req.write_text(_request(b"\x01" * 39))
    assert fn.main(["prove", str(req), "--function", "rotate", "--build-dir", str(b), "--output", str(tmp_path / "o4.json")], OracleBackend) == 2
    assert not (tmp_path / "o4.json").exists()


def test_header_range_256_runs_the_whole_dag(tmp_path):
    """header_range_256 = 32 map + 31 reduce + 1 outer = 64 proofs (tiny stand-in circuits here)."""
    b = tmp_path / "build"
    args = ["--function", "header_range_256", "--build-dir", str(b)]
    assert fn.main(["build", *args, "--map-log-n", "4", "--reduce-log-n", "3", "--outer-log-n", "5"], OracleBackend) == 0
    req = tmp_path / "input.json"
    req.write_text(_request(_header_range_input()))
    backend = OracleBackend()
    proof, out, stats = fn.prove("header_range_256", fn.parse_request(req.read_text(), "header_range_256"), b, backend)
    assert stats["proofs"] == 64 and len(out) == 96
    vc = vx.ParsedCircuit((b / "header_range_256.outer.verifier.vxcircuit").read_bytes())
    vx.verify_standalone(vc.desc_ptr, vc.cap, proof)
    proof2, out2, _ = fn.prove("header_range_256", _header_range_input(target=272535), b, OracleBackend())
    assert out2 != out and proof2 != proof

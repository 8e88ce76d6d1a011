"""Worker for tests/test_gpu_sharded.py: ONE proof split across WORLD_SIZE processes (vx_prove_sharded) with the
exchanges on torch.distributed.  All ranks share GPU 0 (the test boxes have one GPU), so the backend is gloo and
vectorx_amd.sharded.TorchAllGather stages through host memory; on a real node the same code runs with backend nccl."""
import hashlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402,F401

import vectorx_amd as vx  # noqa: E402
from vectorx_amd import dist_harness as H  # noqa: E402
from vectorx_amd import sharded  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402


def main():
    rank, world, _ = H.env_rank()
    dist = H.init(os.environ.get("VX_TEST_BACKEND", "gloo"), 0)
    degree_bits = int(os.environ.get("VX_TEST_DEGREE_BITS", "9"))
    sc = SynthCircuit(degree_bits, seed=4242, poseidon_percent=45, flags=int(os.environ.get("VX_TEST_FLAGS", "0")))
    sc.desc.pow_bits = 8
    ctx = vx.Context(0)
    circuit = vx.Circuit(ctx, sc.desc_ptr)
    w = sc.witness()
    single = circuit.prove(w)
    ag = sharded.TorchAllGather(ctx, dist, torch.device("cuda", 0)) if dist is not None else None
    proof = circuit.prove_sharded(w, rank, world, ag) if dist is not None else single
    mine = [rank, proof == single, hashlib.sha256(proof).hexdigest(), ag.calls if ag else 0, ag.bytes if ag else 0]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    else:
        gathered = [mine]
    circuit.free()
    ctx.close()
    if rank == 0:
        print(json.dumps({"world": world, "results": gathered, "proof_bytes": len(proof)}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

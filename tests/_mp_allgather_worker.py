"""Worker for tests/test_multiproc.py::test_torch_allgather_gloo — the host-side all-gather that vx_prove_sharded
calls back into (vectorx_amd.sharded.TorchAllGather), on gloo with a stand-in for device memory (CPU only)."""
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from vectorx_amd import dist_harness as H  # noqa: E402
from vectorx_amd import sharded  # noqa: E402


class HostMemoryCtx:
    """download/upload with the signature of vectorx_amd.Context, on plain host addresses."""

    def download(self, ptr, nbytes):
        return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=np.uint64).copy()

    def upload(self, ptr, host):
        host = np.ascontiguousarray(host)
        ctypes.memmove(ptr, host.ctypes.data, host.nbytes)


def main():
    rank, world, local_rank = H.env_rank()
    dist = H.init("gloo", local_rank)
    ag = sharded.TorchAllGather(HostMemoryCtx(), dist)
    ok = True
    for words in (4, 64, 1000):
        buf = np.full(world * words, 0xDEAD, dtype=np.uint64)
        buf[rank * words:(rank + 1) * words] = np.arange(words, dtype=np.uint64) + 1000 * rank
        ag(buf.ctypes.data, words * 8)
        want = np.concatenate([np.arange(words, dtype=np.uint64) + 1000 * r for r in range(world)])
        ok = ok and bool((buf == want).all())
    mine = [rank, ok, ag.calls, ag.bytes]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        print(json.dumps({"world": world, "results": gathered}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

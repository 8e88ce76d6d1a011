"""Per-rank cost of ONE header_range_512-sized proof split across G ranks (vx_prove_sharded), measured on a ONE-GPU box.

The G ranks run as host threads on the same device, serialised by a turnstile: a rank computes only while it holds the
lock and gives it up inside every all-gather, so each rank's busy time is what it would take with the GPU to itself —
i.e. the per-rank COMPUTE latency of the sharded proof on a real G-GPU node.  The exchanges themselves (device-local
copies here) are not representative; their volume is reported instead so the xGMI time can be bounded.

usage: python tools/sharded_prove_bench.py [degree_bits] [worlds, e.g. 1,2,4,8] [dev]      ("dev": witness already in HBM)
"""
import os
os.environ.setdefault("VX_NO_WARM_ON_LOAD", "1")   # G ranks share ONE device here: no rank may cache a full-size unsharded working set
import ctypes
import json
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vectorx_amd as vx  # noqa: E402
from vectorx_amd.synth import SynthCircuit  # noqa: E402


def run(sc, w, world, reps=2, dev_witness=False):
    L = vx.lib()
    ctxs = [vx.Context(0) for _ in range(world)]
    circuits = [vx.Circuit(c, sc.desc_ptr) for c in ctxs]
    g = ctypes.c_void_p()
    vx._chk(L.vx_group_create(world, ctypes.byref(g)))
    members = []
    for r in range(world):
        m = ctypes.c_void_p()
        vx._chk(L.vx_group_join(g, r, ctxs[r]._h, ctypes.byref(m)))
        members.append(m)
    turn = threading.Lock()
    busy = [[0.0] * world for _ in range(reps)]
    xbytes = [0] * world
    xcalls = [0] * world
    proofs = [None] * world
    ctxs[0].prof_enable(True)
    d_w = None
    if dev_witness:            # one copy in HBM, visible to every context of this GPU
        d_w = ctxs[0].alloc(w.nbytes)
        ctxs[0].upload(d_w, w)

    def rank_main(r):
        for rep in range(reps):
            if rep == reps - 1 and r == 0:
                ctxs[0].prof_reset()
            state = {"t": 0.0}

            def ag(ptr, nbytes):
                busy[rep][r] += time.perf_counter() - state["t"]
                turn.release()
                if rep == 0:
                    xbytes[r] += nbytes * (world - 1)
                    xcalls[r] += 1
                rc = L.vx_group_allgather(members[r], ptr, nbytes)
                turn.acquire()
                state["t"] = time.perf_counter()
                if rc:
                    raise RuntimeError("allgather failed")

            turn.acquire()
            state["t"] = time.perf_counter()
            try:
                proofs[r] = circuits[r].prove_sharded(None if dev_witness else w, r, world, ag if world > 1 else None, dev_ptr=d_w)
            finally:
                busy[rep][r] += time.perf_counter() - state["t"]
                turn.release()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    prof = ctxs[0].prof()
    if d_w is not None:
        ctxs[0].free(d_w)
    L.vx_group_destroy(g)
    for c in circuits:
        c.free()
    for c in ctxs:
        c.close()
    return proofs, busy[-1], xbytes, xcalls, prof


def main():
    db = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
    dev = len(sys.argv) > 3 and sys.argv[3] == "dev"
    sc = SynthCircuit(db, seed=0, poseidon_percent=50)
    w = sc.witness()
    ref = None
    for world in worlds:
        proofs, busy, xb, xc, prof = run(sc, w, world, dev_witness=dev)
        if ref is None:
            ref = proofs[0]
        same = all(p == ref for p in proofs)
        stages = {k: round(v["ms"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        print(json.dumps({"degree_bits": db, "world": world, "witness": "hbm" if dev else "host", "identical_to_first": same,
                          "per_rank_busy_ms": [round(b * 1e3, 1) for b in busy], "max_busy_ms": round(max(busy) * 1e3, 1),
                          "exchange_calls": xc[0], "exchange_bytes_in_per_rank": xb[0], "rank0_stage_ms": stages}), flush=True)


if __name__ == "__main__":
    main()

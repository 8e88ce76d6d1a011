"""Whole-proof workload for bench.py: one step = one vx_prove of a header_range_512-shaped synthetic circuit
(BASELINE.json configs[2]: 2^21 rows, 135 wires, blow-up 8, full FRI) with the witness already in HBM."""
import sys
import time
from pathlib import Path

import numpy as np

if str(Path(__file__).resolve().parent) not in sys.path:      # run as a script (the CPU baseline's child process)
    sys.path.insert(0, str(Path(__file__).resolve().parent))


def make_step(ctx, args, rank, dist=None, device=None):
    import vectorx_amd as vx
    from vectorx_amd.synth import SynthCircuit

    sharded_mode = getattr(args, "mode", "throughput") == "sharded" and dist is not None
    # throughput mode: every rank proves its OWN witness; sharded mode: all ranks work on the SAME proof
    # ONE circuit on every rank (so that the sharded leg after a throughput run can put all ranks on one proof); in throughput
    # mode each rank proves its own witness of it (rank 0: witness seed = circuit seed, the N = 1 workload)
    shape = {"flags": getattr(args, "circuit_flags", 0)}
    if getattr(args, "recursion_mix", False):         # profiling runs: the DAG's circuit family (recursive verifier's gate set, declared mix) as THE workload
        from vectorx_amd.mapreduce import circuit_shape
        shape = circuit_shape(True)
    sc = SynthCircuit(args.log_n, seed=0x5EED0000, poseidon_percent=args.poseidon_percent,
                      witness_seed=0x5EED0000 + (0 if sharded_mode else rank), **shape)
    counts = sc.row_counts()
    circuit = vx.Circuit(ctx, sc.desc_ptr)   # constants_sigmas commitment stays resident (per-circuit, not per-proof)
    w = sc.witness()
    d_w = ctx.alloc(w.nbytes)
    ctx.upload(d_w, w)
    del w
    sc.release_host_buffers(witness=True, preprocessed=True)
    state = {"proof": None}

    if sharded_mode:
        from vectorx_amd.sharded import TorchAllGather
        ag = TorchAllGather(ctx, dist, device)
        _LEG["allgather"] = ag
        world = dist.get_world_size()

        def step():
            state["proof"] = circuit.prove_sharded(None, rank, world, ag, dev_ptr=d_w)
            return state["proof"]
    else:
        def step():
            state["proof"] = circuit.prove(dev_ptr=d_w)
            return state["proof"]

    _LEG["circuit"], _LEG["d_w"] = circuit, d_w
    metric = "header_range_512 proofs/sec"
    unit = "proofs/sec"
    pct = lambda k: round(100.0 * counts[k] / (1 << args.log_n))   # noqa: E731
    wl = (f"header_range_512 stand-in (BASELINE configs[2]): one plonky2 proof, n=2^{args.log_n} rows x 135 wires, blow-up 8, cap height 4, "
          f"FRI arity 16 / 28 queries / 16 PoW bits; synthetic circuit {pct('poseidon')}% PoseidonGate / {pct('arithmetic')}% ArithmeticGate rows"
          + (f", circuit flags {args.circuit_flags}" if getattr(args, "circuit_flags", 0) else "")
          + (" — NOT the headline mix: the recursive verifier's gate set in its declared row mix (--recursion-mix)" if getattr(args, "recursion_mix", False) else ""))
    def cleanup():
        circuit.free()
        ctx.free(d_w)
        sc.free()

    return step, metric, unit, wl, cleanup


_LEG = {}


def host_witness_leg(ctx, args, sync):
    """K proofs from a PINNED HOST copy of the same witness (vx_prove uploads it in column blocks behind the first
    transforms): the PCIe-inclusive rate.  Returns a small dict for the bench line; proofs must equal the HBM-resident ones."""
    circuit, d_w = _LEG["circuit"], _LEG["d_w"]
    n = 1 << args.log_n
    host = ctx.host_alloc((135, n))
    ctx.download_into(host, d_w)
    ref = circuit.prove(dev_ptr=d_w)
    p = circuit.prove(host)          # warm-up + identity check
    assert p == ref, "proof from the host witness differs from the proof from the HBM-resident witness"
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        circuit.prove(host)
    sync()
    dt = time.perf_counter() - t0
    ctx.host_free(host)
    return {"value": args.steps / dt, "unit": "proofs/sec", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps,
            "what": "same circuit and witness, witness in page-locked HOST memory when each step starts (2.27 GB over PCIe per proof at "
                    "n = 2^21, hidden behind the first transforms and the leaf hashing, which runs in carried-state launches); proofs byte-identical to the HBM-resident run"}


def dag_leg(ctx, local_rank, in_flight=3):
    """The REAL unit of work of the metric, outside the contract's timed region: one header_range_512 = 64 map + 63 reduce
    + 1 outer plonky2 proofs (/root/reference/circuits/builder/subchain_verification.rs:72-78, header_range.rs:71-88)
    scheduled layer by layer on this GPU with `in_flight` proofs in flight (vectorx_amd/mapreduce.py).  Stand-in circuit
    sizes 2^18 / 2^16 / 2^19 rows (the real degrees need the Rust builder); witnesses resident in HBM before the clock starts."""
    import vectorx_amd as vx
    from vectorx_amd import mapreduce as mr
    t_setup = time.perf_counter()
    lanes = [vx.Context(local_rank) for _ in range(in_flight - 1)]
    spec = mr.DagSpec(64, 18, 16, 19)
    provers = {}

    def make(kind, log_n, jobs):        # one prover per circuit kind, kept across the two passes
        if kind not in provers:
            provers[kind] = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, extra_lanes=lanes, distinct_witnesses=4, recursion=spec.recursion)
        return provers[kind]

    def sync():
        ctx.sync()
        for l in lanes:
            l.sync()

    try:
        # three passes over the same DAG: layered, layered again, and the dependency-driven schedule (no layer barriers; one process
        # only) — same proofs, same root.  The FIRST pass is the headline (round 4: vx_circuit_create rehearses one proof, so the pool
        # holds every buffer shape before the clock starts and the first pass no longer pays first-use allocations); all are listed.
        schedules = ["layer barriers", "layer barriers", "dependency-driven"]
        runs = [mr.run_dag(spec, make, None, sync, in_flight=in_flight, barriers=(sch == "layer barriers")) for sch in schedules]
        assert runs[0]["root"] == runs[1]["root"] == runs[2]["root"]
        res = runs[0]
    finally:
        for p in provers.values():
            p.free()
        for l in lanes:
            l.close()
    secs = res["seconds"]
    return {"header_range_512_per_sec": 1.0 / secs, "dag_seconds": secs, "plonky2_proofs": res["proofs"],
            "plonky2_proofs_per_sec": res["proofs"] / secs, "in_flight_per_gpu": in_flight, "schedule": schedules[0], "reported_pass": "first",
            "dag_seconds_all_passes": [[sch, round(r["seconds"], 4)] for sch, r in zip(schedules, runs)],
            "per_layer_ms": [(l["kind"], l["jobs"], round(l["ms"], 1)) for l in res["per_layer"]],
            "non_map_layers_ms": round(sum(l["ms"] for l in res["per_layer"] if l["kind"] != "map"), 1),
            "setup_seconds_untimed": round(time.perf_counter() - t_setup - sum(r["seconds"] for r in runs), 2), "root": res["root"][:32].hex(),
            "what": "64 map (2^18 rows) + 63 reduce (2^16) + 1 outer (2^19) plonky2 proofs, the FIRST pass with layer barriers (per_layer_ms add up to "
                    "dag_seconds; the dependency-driven pass is listed in dag_seconds_all_passes only), synthetic stand-in circuits; witnesses "
                    "HBM-resident (4 base witnesses per circuit kind, each job's own public inputs patched in: the proving work of 128 distinct "
                    "proofs without 128 CPU witness generations); NOT the contract's timed region"}


def dag_pool_legs(pool, with_starks=True, passes=2):
    """The header_range_512 DAG on a pool of worker PROCESSES per GPU (vectorx_amd/dag_pool.py: started before this process's first GPU
    call, configured here): 64 map + 63 reduce + 1 outer jobs, a job to whichever worker has a free lane, digest + statement records back over a
    pipe.  Two figures: the plonky2 proofs alone, and every job WITH its STARK tables (per-job inputs, traces generated on the GPU
    inside the clock).  The FIRST pass of each is the headline; all passes are listed.
    -> {"dag_header_range_512": {...}, "dag_header_range_512_with_starks": {...}}"""
    t0 = time.perf_counter()
    ready = pool.wait_ready()
    setup_s = time.perf_counter() - t0
    spec, out = pool.spec, {}
    common = {"workers_per_gpu": pool.wpd, "lanes_per_worker": pool.lanes,
              "schedule": "dependency-driven: a job starts when ITS children are proven and goes to the worker with most free lanes; the outer job's STARK "
                          "tables depend on the request only and are proven on worker 0's outer lane while the tree is still being reduced (the pass "
                          "with strict layer barriers and nothing hoisted is listed in dag_seconds_all_passes)",
              "setup_seconds_untimed": round(setup_s, 2),
              "setup_seconds_by_worker": [r["setup_seconds"] for r in ready],
              "setup_is": "worker start-up: circuit builds (synthetic generator + constants/sigmas commitment + one rehearsal proof per lane), "
                          "loading compiled constraint programs, one warm-up proof of every table per lane; NO trace generation for the per-job tables"}

    def record(runs, what):
        res = runs[0]
        secs = res["seconds"]
        return {"header_range_512_per_sec": 1.0 / secs, "dag_seconds": secs, "dag_seconds_all_passes": [[r["schedule"], round(r["seconds"], 4)] for r in runs],
                "plonky2_proofs": res["proofs"], "plonky2_proofs_per_sec": res["proofs"] / secs, **common,
                "lane_seconds_by_kind": {k: round(v, 4) for k, v in sorted(res["split"].items())},
                "per_layer_ms": [(l["kind"], l["jobs"], round(l["ms"], 1)) for l in res["per_layer"]],
                "per_layer_ms_is": "first start to last end of a layer's jobs; layers overlap under the dependency-driven schedule",
                "per_layer_ms_layer_barriers": [(l["kind"], l["jobs"], round(l["ms"], 1)) for l in runs[-1]["per_layer"]],
                "non_map_layers_ms_layer_barriers": round(sum(l["ms"] for l in runs[-1]["per_layer"] if l["kind"] != "map"), 1),
                "jobs_by_worker": res["jobs_by_worker"], "root": res["root"][:32].hex(), "what": what}

    sizes = f"{spec.num_map} map (2^{spec.map_log_n} rows) + {spec.num_map - 1} reduce (2^{spec.reduce_log_n}) + 1 outer (2^{spec.outer_log_n}) plonky2 proofs"
    schedules = ["dependency"] * passes + ["layers"]
    runs = [pool.run(b"bench request", with_tables=False, schedule=sch) for sch in schedules]
    assert len({r["root"] for r in runs}) == 1
    out["dag_header_range_512"] = record(runs, sizes + ", synthetic stand-in circuits; witnesses HBM-resident (4 base witnesses per circuit kind and lane, "
                                         "each job's own public inputs patched in); NOT the contract's timed region")
    try:   # one more pass with HIP-event profiling on every lane: where the 128 proofs' GPU time goes, the quotient by kernel (not a timed pass)
        pool.profile(True)
        pr = pool.run(b"bench request", with_tables=False, schedule="dependency")
        stages = pool.profile_get()
        pool.profile(False)
        top = {k: round(v, 1) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]) if not k.startswith("qgate_") and k not in QUOTIENT_NESTED and v >= 1.0}
        out["dag_header_range_512"]["stage_elapsed_ms_per_dag"] = dict(list(top.items())[:8])
        out["dag_header_range_512"]["quotient_elapsed_by_kernel_ms_per_dag"] = {k[len("quotient_"):]: round(stages[k], 1) for k in QUOTIENT_NESTED if k in stages}
        out["dag_header_range_512"]["profiled_pass_seconds"] = round(pr["seconds"], 4)
    except Exception as e:   # noqa: BLE001 — a diagnostic, never the leg
        out["dag_header_range_512"]["stage_profile_error"] = repr(e)[:160]
    if with_starks:
        t0 = time.perf_counter()
        pool.load_request(b"bench request")          # the request's input (header chain, justification) in the workers' host memory
        load_s = time.perf_counter() - t0
        runs = [pool.run(b"bench request", schedule=sch) for sch in schedules]
        assert len({r["root"] for r in runs}) == 1
        tables = next((r.get("tables") for r in ready if r.get("worker") == 0), None)
        rec = record(runs, sizes + ", EACH JOB WITH ITS STARK TABLES (own AIRs standing in for Curta's chips): map = BLAKE2b over the job's own 8 headers "
                                  "(2240 compressions, 2^16 rows) + SHA-256 over the 14 nodes of their state / data root trees; reduce = SHA-256 over its "
                                  "children's roots; outer = SHA-256 over the authority set commitment chain (300 keys) + the 300 signatures of the precommit "
                                  "verified THROUGH TABLES ONLY as one bus (SHA-512 over R || A || M, 4 batched EdDSA tables running the full program — three of 2^20 rows and the last sized to the 9 signatures left over, 2^17 —, the link "
                                  "table, the verifier's sink).  ONE SYNTHETIC REQUEST (vectorx_amd/header_range.py: 512 headers of 35 840 bytes chained by their "
                                  "hashes, real Ed25519 authorities, a signed precommit), resident in host memory when the clock starts; every job hashes ITS "
                                  "part of it and its children's statements, TRACES GENERATED ON THE GPU INSIDE THE CLOCK (vx_trace_*: lane-seconds "
                                  "`trace_generation`), and states what the reference's circuit asserts (linked headers, merged roots, the justification) from "
                                  "the tables' digests: the outer job's statement is the function's 96 OUTPUT BYTES (`output`), equal to the host computation "
                                  "over the same request.  The STARK proofs and the statement are part of a job's digest; NOT the contract's timed region")
        rec["tables"] = tables
        rec["request_load_seconds_untimed"] = round(load_s, 2)
        rec.update(statement_record(pool.cfg, spec, b"bench request", runs[0]["root"]))
        rec["stark_proofs"] = spec.num_map * 2 + (spec.num_map - 1) + 2 + ((tables or {}).get("eddsa_outer", {}).get("tables", 0)) \
            + (2 if "signature_bus" in (tables or {}) else 0)
        out["dag_header_range_512_with_starks"] = rec
    return out


def statement_record(cfg, spec, seed: bytes, root: bytes) -> dict:
    """The function I/O of a DAG run with per-job tables: the request's 80 input bytes, the 96 output bytes the outer job stated (the
    tail of the root record) and whether they equal the host computation over the same request (hashlib + avail_codec alone)."""
    from vectorx_amd import dag_tables
    from vectorx_amd import header_range as hr
    if cfg.get("table_mode") != "per_job" or len(root) <= 32:
        return {}
    req = hr.cached_request(seed, **dag_tables.request_shape(cfg["small_tables"], spec.num_map, cfg.get("num_headers")))
    return {"input": req.input_bytes.hex(), "output": root[32:].hex(), "output_equals_host_computation": root[32:] == hr.expected_output(req),
            "output_is": "abi.encode(target_header_hash, state_root_commitment, data_root_commitment) — /root/reference/circuits/header_range.rs:56-58"}


def guarded_collective_leg(dist, fn):
    """Run a leg in which EVERY rank takes part; an exception on any rank becomes {"error": ...} in the line instead of
    costing it.  (A rank that fails inside a collective can still leave its peers waiting: the backend's own timeout ends that.)"""
    try:
        res, err = fn(), None
    except Exception as e:   # noqa: BLE001
        res, err = None, repr(e)
    errs = [None] * dist.get_world_size()
    dist.all_gather_object(errs, err)
    bad = {r: e for r, e in enumerate(errs) if e}
    return {"error": bad} if bad else res


def _max_over_ranks(dist, device, x: float) -> float:
    import torch
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _broadcast_device_buffer(ctx, dist, device, dptr, nbytes, src=0):
    """rank `src`'s device buffer to every rank: in place over RCCL, through host memory in pieces on any other backend"""
    import torch
    from vectorx_amd.sharded import _DeviceView
    if dist.get_backend() == "nccl":
        t = torch.as_tensor(_DeviceView(dptr, nbytes), device=device)
        dist.broadcast(t, src=src)
        torch.cuda.synchronize(device)
        return
    piece = 1 << 28
    for off in range(0, nbytes, piece):
        m = min(piece, nbytes - off)
        h = torch.from_numpy(ctx.download(dptr + off, m).view(np.uint8).copy())
        dist.broadcast(h, src=src)
        if dist.get_rank() != src:
            ctx.upload(dptr + off, h.numpy())


def sharded_one_proof_leg(ctx, args, rank, world, dist, device, sync):
    """BASELINE.json configs[3] AFTER the timed region of an N > 1 run: ONE proof of the bench's circuit (n = 2^log_n) over all N
    ranks (`vx_prove_sharded`: coset split, every exchange an all-gather the host supplies — RCCL over xGMI under backend nccl),
    witness resident in HBM on every rank.  Reports ms per proof (max over ranks), the exchanges the library asked for, the bytes
    a rank received and the host-clock time rank 0 spent inside the exchanges (incl. waiting for the slowest rank)."""
    from vectorx_amd.sharded import TorchAllGather
    circuit, d_w = _LEG["circuit"], _LEG["d_w"]
    if world & (world - 1) or world > 8:
        return {"skipped": f"the coset split needs a power-of-two rank count <= 8, got {world}"}
    nbytes = 135 * (8 << args.log_n)
    _broadcast_device_buffer(ctx, dist, device, d_w, nbytes)          # every rank works on rank 0's witness
    ag = TorchAllGather(ctx, dist, device)
    whole = circuit.prove(dev_ptr=d_w) if rank == 0 else None        # the unsharded proof of the same witness, for the identity check
    proof = circuit.prove_sharded(None, rank, world, ag, dev_ptr=d_w)  # warm-up (first-use allocations of the sharded shapes)
    same = [None] * world
    dist.all_gather_object(same, __import__("hashlib").sha256(proof).hexdigest())
    steps = max(1, args.sharded_leg_steps)
    sync()
    dist.barrier()
    ctx.prof_enable(True)
    ctx.prof_reset()
    ag.calls = ag.bytes = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        circuit.prove_sharded(None, rank, world, ag, dev_ptr=d_w)
    sync()
    dist.barrier()
    dt = _max_over_ranks(dist, device, time.perf_counter() - t0)
    prof = ctx.prof()
    ctx.prof_enable(False)
    wait = prof.get("exchange_host_wait", {"ms": 0.0})["ms"] / steps
    return {"ms_per_proof": round(dt / steps * 1e3, 3), "proofs_per_sec": steps / dt, "ranks": world, "steps": steps,
            "log_n": args.log_n, "scaling": "strong", "backend": dist.get_backend(),
            "allgather_calls_per_proof": ag.calls / steps, "inbound_bytes_per_rank_per_proof": ag.bytes / steps,
            "exchange_host_wait_ms_rank0": round(wait, 3), "exchange_host_wait_ms": round(wait, 3),
            "rank0_stage_ms": {k: round(v["ms"] / steps, 3) for k, v in prof.items() if v["ms"] / steps >= 0.05},
            "all_ranks_returned_the_same_proof": len(set(same)) == 1,
            "byte_identical_to_unsharded_vx_prove": (proof == whole) if rank == 0 else None,
            "identical_to_unsharded": (proof == whole) if rank == 0 else None, "rccl_ranks": world,
            "what": "one proof coset-sharded over all ranks (vx_prove_sharded), witness HBM-resident on every rank; NOT the contract's timed region"}


def dag_leg_ranks(ctx, args, local_rank, dist, device, in_flight=3, with_starks=False):
    """One header_range_512 DAG over ALL ranks (what tools/dag_bench.py does under a launcher): jobs of a layer dealt round-robin,
    `in_flight` proofs in flight per GPU, an all-gather of 32-byte proof digests at every layer barrier.  Runs twice; the FIRST pass
    is the headline (circuits pre-size their pools at load), the second is listed."""
    import vectorx_amd as vx
    from vectorx_amd import mapreduce as mr
    t_setup = time.perf_counter()
    num_map, lm, lr, lo = (int(x) for x in args.dag_spec.split(","))
    spec = mr.DagSpec(num_map, lm, lr, lo)
    lanes = [vx.Context(local_rank) for _ in range(in_flight - 1)]
    provers, per_kind, tables, split = {}, {}, [], ({} if with_starks else None)
    if with_starks:
        # every rank proves map and reduce jobs; the outer job (and its tables) belongs to the rank job 0 of the last layer is dealt to: rank 0
        kinds = ("map", "reduce", "outer") if dist.get_rank() == 0 else ("map", "reduce")
        per_kind, tables, _setup = dag_stark_tables(ctx, kinds=kinds, small=bool(getattr(args, "dag_starks_small", False)),
                                                    mode=getattr(args, "dag_table_mode", "per_job"), lanes=[ctx] + lanes,
                                                    outer_lanes=[ctx], num_map=num_map)     # layer barriers: the outer job is alone and runs on lane 0

    def make(kind, log_n, jobs):
        if kind not in provers:
            provers[kind] = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, extra_lanes=lanes, distinct_witnesses=4, recursion=spec.recursion,
                                         starks=per_kind.get(kind, ()), split=split)
        return provers[kind]

    def sync():
        ctx.sync()
        for l in lanes:
            l.sync()

    try:
        runs = []
        for _ in range(2):
            if split is not None:
                split.clear()
            r = mr.run_dag(spec, make, dist, sync, in_flight=in_flight)
            r["seconds"] = _max_over_ranks(dist, device, r["seconds"])
            r["split"] = dict(split or {})
            runs.append(r)
        assert runs[0]["root"] == runs[1]["root"]
    finally:
        for p in provers.values():
            p.free()
        for t in tables:
            t.free()
        for l in lanes:
            l.close()
    res = runs[0]
    secs = res["seconds"]
    return {"header_range_512_per_sec": 1.0 / secs, "dag_seconds": secs, "ranks": dist.get_world_size(), "plonky2_proofs": res["proofs"],
            "plonky2_proofs_per_sec": res["proofs"] / secs, "in_flight_per_gpu": in_flight, "schedule": "layer barriers", "scaling": "strong",
            "dag_seconds_all_passes": [round(r["seconds"], 4) for r in runs],
            "per_layer_ms": [(l["kind"], l["jobs"], round(l["ms"], 1)) for l in res["per_layer"]],
            "setup_seconds_untimed": round(time.perf_counter() - t_setup - sum(r["seconds"] for r in runs), 2), "root": res["root"][:32].hex(),
            "backend": dist.get_backend(), "with_stark_tables": bool(with_starks),
            **(statement_record({"table_mode": getattr(args, "dag_table_mode", "per_job"), "small_tables": bool(getattr(args, "dag_starks_small", False))},
                                spec, b"", res["root"]) if with_starks else {}),
            "rank0_lane_seconds_by_kind": {k: round(v, 4) for k, v in sorted(res["split"].items())} if with_starks else None,
            "what": f"{num_map} map (2^{lm} rows) + {num_map - 1} reduce (2^{lr}) + 1 outer (2^{lo}) plonky2 proofs over all ranks"
                    + (", EACH WITH ITS STARK TABLES (dag_stark_tables: BLAKE2b + SHA-256 per map job, SHA-256 per reduce job, SHA-256 / SHA-512 / "
                       "batched EdDSA for the outer job)" if with_starks else "")
                    + ": layer jobs round-robin, digests all-gathered at each layer barrier, synthetic stand-in circuits, witnesses and traces "
                      "HBM-resident; NOT the contract's timed region"}


def dag_stark_tables(ctx, kinds=("map", "reduce", "outer"), small=False, mode="resident", lanes=None, outer_lanes=None, num_map=64, request_seed=b""):
    """vectorx_amd.dag_tables.build (kept under this name for tools/) + the request of `request_seed` derived in this process before any
    clock starts (per-job tables: the header chain; with the outer job also the signed justification)"""
    from vectorx_amd import dag_tables
    out = dag_tables.build(ctx, kinds=kinds, small=small, mode=mode, lanes=lanes, outer_lanes=outer_lanes, num_map=num_map)
    if mode == "per_job" and request_seed is not None:
        dag_tables.preload_request(request_seed, dag_tables.request_shape(small, num_map), outer="outer" in kinds)
    return out


def dag_with_starks_leg(ctx, local_rank, in_flight=None, table_mode="per_job"):
    """the real job mix — every plonky2 proof of the header_range_512 DAG WITH the STARK tables its circuit embeds (see
    dag_stark_tables), on one GPU, `in_flight` jobs in flight.  Reports the DAG's wall time and, per kind of work, the LANE-seconds spent
    in it (the lanes overlap, so the kinds add up to about in_flight x the wall time)."""
    import os

    import vectorx_amd as vx
    from vectorx_amd import mapreduce as mr
    if in_flight is None:
        in_flight = int(os.environ.get("VX_DAG_STARKS_IN_FLIGHT", "3"))
    t_setup = time.perf_counter()
    lanes = [vx.Context(local_rank) for _ in range(in_flight - 1)]
    # every lane proves every table once (untimed, inside build): its pool then holds the STARK shapes too, like vx_circuit_warm does
    # for the plonky2 shapes — the first timed pass allocates nothing
    # (the outer job is alone in its layer and runs on lane 0: only that lane holds the signature bus's buffers — 21 GB of traces, 42 GB
    # of sessions while it proves; on every lane they would not fit next to the circuits)
    per_kind, tables, setup = dag_stark_tables(ctx, mode=table_mode, lanes=[ctx] + lanes, outer_lanes=[ctx])
    spec = mr.DagSpec(64, 18, 16, 19)
    provers, split = {}, {}

    def make(kind, log_n, jobs):
        if kind not in provers:
            provers[kind] = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, extra_lanes=lanes, distinct_witnesses=4, recursion=spec.recursion,
                                         starks=per_kind[kind], split=split)
        return provers[kind]

    def sync():
        ctx.sync()
        for l in lanes:
            l.sync()

    try:
        runs = []
        for _ in range(2):
            split.clear()
            r = mr.run_dag(spec, make, None, sync, in_flight=in_flight)
            r["split"] = dict(split)
            runs.append(r)
        assert runs[0]["root"] == runs[1]["root"]
    finally:
        for p in provers.values():
            p.free()
        for t in tables:
            t.free()
        for l in lanes:
            l.close()
    res = runs[0]                       # the FIRST pass is the headline
    secs = res["seconds"]
    stark_proofs = 64 * 2 + 63 * 1 + 2 + setup["eddsa_outer"]["tables"] + (2 if "signature_bus" in setup else 0)
    what_tables = ("EVERY JOB ITS OWN TABLES: a map job's 8 headers / tree nodes, a reduce job's merge nodes, the outer job's authority set and signed "
                   "messages are derived from the request seed and the job's position, and the traces are generated on the GPU (vx_trace_*) INSIDE "
                   "the clock (lane-seconds `trace_generation`), the four batched EdDSA tables included"
                   if table_mode == "per_job" else
                   "every trace, witness and second-round column resident in HBM before the clock starts (one trace per table kind, proven once per job)")
    return {"header_range_512_per_sec": 1.0 / secs, "dag_seconds": secs, "dag_seconds_all_passes": [round(r["seconds"], 4) for r in runs],
            "plonky2_proofs": res["proofs"], "stark_proofs": stark_proofs, "in_flight_per_gpu": in_flight, "schedule": "layer barriers",
            "lane_seconds_by_kind": {k: round(v, 4) for k, v in sorted(res["split"].items())},
            "per_layer_ms": [(l["kind"], l["jobs"], round(l["ms"], 1)) for l in res["per_layer"]],
            "tables": setup, "setup_seconds_untimed": round(time.perf_counter() - t_setup - sum(r["seconds"] for r in runs), 2),
            "root": res["root"][:32].hex(), **statement_record({"table_mode": table_mode, "small_tables": False}, spec, b"", res["root"]),
            "what": "64 map jobs = plonky2 2^18 + BLAKE2b table (2240 compressions: 2^16 rows of the byte / XOR-lookup table) + SHA-256 table 2^11; 63 reduce jobs = plonky2 2^16 + SHA-256 "
                    "table 2^9; outer = plonky2 2^19 + SHA-256 chain 2^16 + SHA-512 2^16 + 3 batched EdDSA tables 2^20 + one 2^17 (303 signature slots for 300 "
                    "signatures); own AIRs standing in for Curta's chips, synthetic stand-in circuits; " + what_tables + "; the STARK proofs are part of a "
                    "job's digest; NOT the contract's timed region"}


def rotate_leg(ctx, local_rank, log_n=19, small=False):
    """The OTHER function of the hot path (/root/reference/circuits/rotate.rs:80-109, bin/rotate.rs): one rotate request = ONE plonky2
    proof (stand-in circuit, 2^19 rows) + its tables over a synthetic request (an epoch end header announcing 300 new authorities,
    justified by the 300 current ones): BLAKE2b over the header, SHA-256 over both authority set commitment chains, the 300 signatures
    through the signature bus — traces generated on the GPU inside the clock — and the statement: the new authority set's hash, equal to
    the host computation (avail_codec.rotate_output).  Outside the contract's timed region."""
    from vectorx_amd import dag_tables
    from vectorx_amd import header_range as hr
    from vectorx_amd import mapreduce as mr
    t_setup = time.perf_counter()
    import vectorx_amd as vx
    bus_lane = vx.Context(local_rank)          # the signature bus is proven ahead, next to the plonky2 proof and the hash tables
    per_kind, tables, setup = dag_tables.build_rotate(ctx, [ctx], small=small, bus_lane=bus_lane)
    ahead = setup.pop("ahead")
    prover = mr.GpuProver(ctx, "rotate", log_n, [(0, 0)], 50, distinct_witnesses=1, starks=per_kind["rotate"])
    try:
        seeds = [b"bench rotate 1", b"bench rotate 2"]
        reqs = [hr.cached_request(sd, rotate=True, **dag_tables.rotate_shape(small)) for sd in seeds]
        for r in reqs:
            dag_tables._signature_inputs(r)          # the request's input, signatures included, in host memory before the clock
        setup_s = time.perf_counter() - t_setup
        runs = []
        for sd in seeds:
            spent = []
            res = mr.prove_rotate(prover, sd, spent_out=spent, ahead=ahead)
            ctx.sync()
            split = {}
            for k, v in spent:
                split[k] = split.get(k, 0.0) + v
            res["split"] = split
            runs.append(res)
    finally:
        prover.free()
        for t in tables:
            t.free()
        bus_lane.close()
    ok = [r["output"] == hr.expected_rotate_output(q) for r, q in zip(runs, reqs)]
    res = runs[0]
    return {"rotate_per_sec": 1.0 / res["seconds"], "seconds": round(res["seconds"], 4), "seconds_all_passes": [round(r["seconds"], 4) for r in runs],
            "seconds_by_kind": {k: round(v, 4) for k, v in sorted(res["split"].items())}, "input": reqs[0].input_bytes.hex(), "output": res["output"].hex(),
            "output_equals_host_computation": all(ok), "stark_proofs": 2 + 3 + setup["signature_bus"]["eddsa_tables"], "tables": setup,
            "setup_seconds_untimed": round(setup_s, 2),
            "what": f"ONE rotate request: plonky2 2^{log_n} (synthetic stand-in circuit) + BLAKE2b table over the epoch end header (2^16 rows) + SHA-256 table over the "
                    "current and the new authority set commitment chains (1198 compressions, 2^17 rows) + the justification's 300 signatures as one bus "
                    "(SHA-512 2^16 + 3 EdDSA full 2^20 + 1 EdDSA full 2^17 for the 9 signatures left over + link + sink) proven AHEAD on a lane of its own next to the plonky2 proof and the hash tables (seconds_by_kind "
                    "adds up to more than `seconds`), traces generated on the GPU inside the clock; the statement = the new "
                    "authority set hash (bytes32: /root/reference/circuits/rotate.rs:108); NOT the contract's timed region"}


def chip_leg(ctx):
    """SURVEY §8 f-3 outside the contract's timed region: the STARK path on the four chip tables the header_range jobs prove — SHA-256,
    SHA-512, BLAKE2b (bytes + XOR lookup) and the batched EdDSA equations, own AIRs standing in for Curta's chips
    (/root/reference/circuits/builder/header.rs:18, justification.rs:140-156, 237) — ONE table each, lone proofs.  Every figure comes
    through stark_chips.ResidentTable: the second-round columns are computed ON THE GPU inside every timed proof (round 4's leg cached a
    host computation).  The hash tables' traces are generated on the GPU (vx_trace_*), timed separately; the EdDSA trace comes from the
    host generator.  Compiled constraint programs come from .jit_cache/ when __graft_entry__.build() filled it."""
    import os
    from pathlib import Path

    from vectorx_amd import blake2b_bytes_air, eddsa_air, sha256_air, sha512_air, stark_chips
    cache = Path(__file__).resolve().parent / ".jit_cache"
    if "VX_JIT_CACHE_DIR" not in os.environ and cache.is_dir():
        os.environ["VX_JIT_CACHE_DIR"] = str(cache)
    out = {"what": "own AIRs, not Curta's; rate_bits 1, 84 queries, 16 PoW bits; lone proofs, trace resident in HBM, second-round columns computed "
                   "on the GPU inside every proof, starky's transcript order (the opening set absorbed element by element; "
                   "`ms_per_proof_openings_digest` = the same table under this library's tree-hash variant VX_STARK_OPENINGS_DIGEST); NOT the contract's timed region",
           "openings_digest": 0}
    rng = np.random.default_rng(5)
    cases = [("sha256", "sha256", sha256_air, 13, 60, 64), ("sha512", "sha512", sha512_air, 13, 48, 117),
             ("blake2b_bytes", "blake2b", blake2b_bytes_air, 16, 8, 128 * 280)]
    for name, which, air, log_n, nmsg, mlen in cases:
        msgs = [rng.integers(0, 256, size=mlen, dtype=np.uint8).tobytes() for _ in range(nmsg)]
        stark = air.make_stark(log_n)
        ncols = stark.desc.num_columns
        d = ctx.alloc(ncols * (1 << log_n) * 8)
        ctx.trace_hash_table(which, log_n, msgs, d)            # warm
        t0 = time.perf_counter()
        pis, _ = ctx.trace_hash_table(which, log_n, msgs, d)
        gen_ms = (time.perf_counter() - t0) * 1e3
        r = stark_chips.bench_table(ctx, stark, None, pis, name, steps=3, warmup=1, d_trace=d)
        out[name] = {"ms_per_proof": round(r["ms_per_proof"], 3), "rows_log2": log_n, "columns": r["columns"], "messages": nmsg,
                     "trace_generation_ms_gpu": round(gen_ms, 3), "trace_GB": round(ncols * (1 << log_n) * 8 / 1e9, 3),
                     "stage_ms_per_proof": r["stage_ms_per_proof"], "proof_bytes": r["proof_bytes"], "evaluator": r["evaluator"]}
        d2 = ctx.alloc(ncols * (1 << log_n) * 8)               # (a ResidentTable owns and frees the trace it is given)
        ctx.trace_hash_table(which, log_n, msgs, d2)
        rd = stark_chips.bench_table(ctx, air.make_stark(log_n, openings_digest=True), None, pis, name, steps=3, warmup=1, d_trace=d2)
        out[name]["ms_per_proof_openings_digest"] = round(rd["ms_per_proof"], 3)
    lay = eddsa_air.Layout(full=True)                          # signatures from their bytes: decompression, digest mod L, S < L in the table
    log_n = 17
    cap = eddsa_air.capacity(lay, log_n)
    sigs, rs = stark_chips.eddsa_signatures_full(cap, 2)
    stark = eddsa_air.make_stark(lay, log_n)
    d = ctx.alloc(lay.N * (1 << log_n) * 8)
    ctx.trace_eddsa_table(log_n, lay.NB, sigs, d, full=True)   # warm
    t0 = time.perf_counter()
    res = ctx.trace_eddsa_table(log_n, lay.NB, sigs, d, full=True)
    gen_ms = (time.perf_counter() - t0) * 1e3
    assert res == rs, "a generated EdDSA instance does not arrive at R"
    r = stark_chips.bench_table(ctx, stark, None, np.zeros(0, dtype=np.uint64), "eddsa", steps=3, warmup=1, d_trace=d)
    out["eddsa"] = {"ms_per_proof": round(r["ms_per_proof"], 3), "rows_log2": log_n, "columns": r["columns"], "signatures": cap, "program": "full (encodings, S, digest in; decompression, digest mod L, S < L inside the table)",
                    "trace_generation_ms_gpu": round(gen_ms, 3), "trace_GB": round(lay.N * (1 << log_n) * 8 / 1e9, 3),
                    "stage_ms_per_proof": r["stage_ms_per_proof"], "proof_bytes": r["proof_bytes"], "evaluator": r["evaluator"]}
    d2 = ctx.alloc(lay.N * (1 << log_n) * 8)
    ctx.trace_eddsa_table(log_n, lay.NB, sigs, d2, full=True)
    rd = stark_chips.bench_table(ctx, eddsa_air.make_stark(lay, log_n, openings_digest=True), None, np.zeros(0, dtype=np.uint64), "eddsa", steps=3, warmup=1, d_trace=d2)
    out["eddsa"]["ms_per_proof_openings_digest"] = round(rd["ms_per_proof"], 3)
    return out


def usable_cores():
    """host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota
    (the GPU box reports 256 logical CPUs but the container is limited by cpu.max)."""
    import math
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(args):
    """oracle (kind "port": CPU restatement of plonky2 v0.2.0, OpenMP over all host cores) proving a bounded
    sample — the same circuit family at 2^cpu_sample_log_n rows — scaled linearly in rows to the bench size."""
    import oracle_lib
    from vectorx_amd.synth import SynthCircuit

    oracle = oracle_lib.load()
    cores = usable_cores()
    oracle.L.vxo_set_num_threads(cores)
    s_log = min(args.cpu_sample_log_n, args.log_n)
    sc = SynthCircuit(s_log, seed=0x5EED0000, poseidon_percent=args.poseidon_percent)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    t0 = time.perf_counter()
    proof, tm = oc.prove(sc.witness(), want_timings=True)
    dt = time.perf_counter() - t0
    assert oc.verify(proof) == ""
    scale = float(1 << (args.log_n - s_log))
    out = {
        "value": 1.0 / (dt * scale), "unit": "proofs/sec", "cores": cores, "kind": "port",
        "sample": f"oracle prove() at 2^{s_log} rows in {dt:.2f} s, scaled x{int(scale)} (linear in rows) to 2^{args.log_n}",
    }
    sc.free()
    out["seconds"] = round(dt, 2)
    out["log_n"] = s_log
    out["stages_s"] = {k: round(v, 2) for k, v in tm.items()}
    return out


QUOTIENT_NESTED = ("quotient_l0_permutation", "quotient_small_native_gates", "quotient_poseidon_gate", "quotient_lookup_terms",
                   "quotient_program_gates_jit", "quotient_program_gates")     # stages bracketed INSIDE quotient_eval


def lone_proof_profile(ctx, log_n: int, recursion: bool, steps: int = 10, per_gate: bool = False) -> dict:
    """One circuit of the DAG's family at 2^log_n rows proven `steps` times on its own: wall ms per proof and the HIP-event stage times, the
    quotient split by kernel (`quotient_by_kernel_ms`); per_gate: the program gates as one kernel each (VX_JIT_FUSED=0 for this circuit
    only) so that every gate has its own time (`quotient_by_gate_ms`, by gate name)."""
    import os
    import vectorx_amd as vx
    from vectorx_amd.mapreduce import circuit_shape
    from vectorx_amd.synth import SynthCircuit
    sc = SynthCircuit(log_n, seed=202, poseidon_percent=50, witness_seed=1, **circuit_shape(recursion))
    old = os.environ.get("VX_JIT_FUSED")
    if per_gate:
        os.environ["VX_JIT_FUSED"] = "0"
    try:
        circuit = vx.Circuit(ctx, sc.desc_ptr)
    finally:
        if per_gate:
            if old is None:
                os.environ.pop("VX_JIT_FUSED", None)
            else:
                os.environ["VX_JIT_FUSED"] = old
    w = sc.witness()
    d_w = ctx.alloc(w.nbytes)
    ctx.upload(d_w, w)
    for _ in range(2):
        circuit.prove(dev_ptr=d_w)
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        circuit.prove(dev_ptr=d_w)
    ctx.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    prof = ctx.prof()
    ctx.prof_enable(False)
    stages = {k: round(v["ms"] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    names = sc.gate_names()
    by_gate = {"+".join(names[int(g)] for g in k[6:].split("+")): v for k, v in stages.items() if k.startswith("qgate_")}
    top = {k: v for k, v in stages.items() if not k.startswith("qgate_") and k not in QUOTIENT_NESTED}
    rec = {"log_n": log_n, "recursion_mix": bool(recursion), "ms_per_proof": round(ms, 3), "kernel_ms": round(sum(top.values()), 3),
           "quotient_eval_ms": stages.get("quotient_eval"),
           "quotient_by_kernel_ms": {k[len("quotient_"):]: stages[k] for k in QUOTIENT_NESTED if k in stages},
           "stage_ms": top}
    if by_gate:
        rec["quotient_by_gate_ms"] = by_gate
    if recursion:
        rec["gate_rows"] = sc.gate_rows()
    circuit.free()
    ctx.free(d_w)
    sc.free()
    return rec


def recursion_profile_leg(ctx, spec=None) -> dict:
    """The DAG's three circuit sizes proven alone with the recursion-shaped gate mix: what a job costs outside the saturated pool, and where
    the quotient's time goes kernel by kernel (`quotient_eval` by kernel; per gate at the map size)."""
    from vectorx_amd import mapreduce as mr
    spec = spec or mr.DagSpec()
    out = {}
    for kind in ("reduce", "map", "outer"):
        r = lone_proof_profile(ctx, spec.log_n(kind), True, steps=10)
        out[kind] = {"log_n": r["log_n"], "ms_per_proof": r["ms_per_proof"], "quotient_eval_ms": r["quotient_eval_ms"],
                     "quotient_by_kernel_ms": r["quotient_by_kernel_ms"], "hash_leaves_ms": r["stage_ms"].get("hash_leaves"),
                     "lde_ms": r["stage_ms"].get("lde"), "merkle_levels_ms": r["stage_ms"].get("merkle_levels")}
        base = lone_proof_profile(ctx, spec.log_n(kind), False, steps=10)
        out[kind]["ms_per_proof_two_gate_stand_in"] = base["ms_per_proof"]
    g = lone_proof_profile(ctx, spec.map_log_n, True, steps=5, per_gate=True)
    out["map_quotient_by_gate_ms_one_kernel_per_gate"] = g.get("quotient_by_gate_ms")
    out["gate_rows_map"] = g.get("gate_rows")
    return out


def cpu_baseline_full(args, timeout_s: float = 900.0) -> dict:
    """The CPU baseline MEASURED at the bench size, no scaling: the oracle proves the bench circuit itself — same generator
    seed, 2^log_n rows — once, on all host cores, in a CHILD process (an out-of-memory kill or a hang there costs this record, never the
    bench line) while the GPU idles.  -> {"value", "seconds", "cores", "log_n", "stages_s", "proof_sha256"}"""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cpu-full-child", str(args.log_n), str(args.poseidon_percent)],
                       capture_output=True, text=True, timeout=timeout_s)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"the CPU child ended with code {r.returncode}: {r.stderr[-300:]}")
    return json.loads(lines[-1])


def _cpu_full_child(log_n: int, pct: int) -> None:
    import hashlib
    import json
    sys.path.insert(0, str(Path(__file__).resolve().parent / "tests"))
    import oracle_lib
    from vectorx_amd.synth import SynthCircuit
    oracle = oracle_lib.load()
    cores = usable_cores()
    oracle.L.vxo_set_num_threads(cores)
    sc = SynthCircuit(log_n, seed=0x5EED0000, poseidon_percent=pct)
    oc = oracle_lib.OracleCircuit(oracle, sc.desc_ptr)
    t0 = time.perf_counter()
    proof, tm = oc.prove(sc.witness(), want_timings=True)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": 1.0 / dt, "unit": "proofs/sec", "seconds": round(dt, 2), "cores": cores, "kind": "port", "log_n": log_n,
                      "stages_s": {k: round(v, 2) for k, v in tm.items()}, "proof_sha256": hashlib.sha256(proof).hexdigest()}), flush=True)


# ---- the line the driver keeps (VERDICT r5 #2b) -------------------------------------------------------------------------------------
# The driver's record holds the last 8 KB of stdout: a 16 KB line of mostly prose loses its numbers.  `compact_line` keeps every NUMBER
# and drops the descriptions, which live — keyed by field name — in profiles/bench_line_glossary.md; the complete line of the same run
# (prose included) goes to a side file.
_KEEP_TEXT = {"metric", "unit", "dtype", "data", "scaling", "workload", "parallelism", "kind", "bound", "dist_backend", "error", "evaluator", "backend"}
_HEX_KEYS = {"root", "input", "output", "proof_sha256", "record"}
_FLAG_TEXT = {"emulated_ranks_on_one_device"}     # a label whose PRESENCE is the information: kept as `true`


def compact_line(d, limit: int = 6000):
    """-> a copy of the bench line that serialises to fewer than `limit` bytes: strings longer than 48 characters are dropped unless
    their key is one of the contract's (metric, config.workload ...), hex fields are cut to 16 characters, floats to 6 significant
    digits; if that is not enough, the least important detail tables go (`dropped_for_size` counts them), never a headline number."""
    import json

    def walk(o, key=None):
        if isinstance(o, dict):
            out = {}
            for k, v in o.items():
                w = walk(v, k)
                if w is not None or v is None:
                    out[k] = w
            return out
        if isinstance(o, (list, tuple)):
            return [walk(v, key) for v in o]
        if isinstance(o, float):
            return float(f"{o:.6g}")
        if isinstance(o, str):
            if key in _HEX_KEYS:
                return o[:16]
            if key in _FLAG_TEXT:
                return True
            if key in _KEEP_TEXT or len(o) <= 48:
                return o
            return None
        return o

    c = walk(d)
    c["glossary"] = "profiles/bench_line_glossary.md"
    # the legs keep their headline numbers; detail tables (per-layer times, per-table shapes, stage splits of the chip proofs ...) are in
    # the complete line only
    keep = {
        "dag": ("dag_seconds", "dag_seconds_all_passes", "lane_seconds_by_kind", "rank0_lane_seconds_by_kind", "jobs_by_worker", "workers_per_gpu",
                "lanes_per_worker", "non_map_layers_ms_layer_barriers", "setup_seconds_untimed", "root", "output", "output_equals_host_computation",
                "stark_proofs", "plonky2_proofs", "ranks", "with_stark_tables", "devices", "error", "backend", "seconds", "stage_elapsed_ms_per_dag",
                "quotient_elapsed_by_kernel_ms_per_dag", "profiled_pass_seconds"),
        "chip": ("ms_per_proof", "ms_per_proof_openings_digest", "rows_log2", "proof_bytes", "trace_generation_ms_gpu"),
        "rotate": ("seconds", "seconds_all_passes", "seconds_by_kind", "output", "output_equals_host_computation", "stark_proofs", "setup_seconds_untimed", "error"),
        "alone": ("log_n", "ms_per_proof", "ms_per_proof_two_gate_stand_in", "quotient_eval_ms", "quotient_by_kernel_ms"),
    }

    def slim(node, names):
        return {k: v for k, v in node.items() if k in names} if isinstance(node, dict) else node

    for leg in ("dag_header_range_512", "dag_header_range_512_with_starks"):
        if isinstance(c.get(leg), dict):
            c[leg] = slim(c[leg], keep["dag"])
    pool = c.get("dag_on_one_pool_over_all_gpus")
    if isinstance(pool, dict):
        c["dag_on_one_pool_over_all_gpus"] = {k: (slim(v, keep["dag"]) if isinstance(v, dict) else v) for k, v in pool.items()}
    if isinstance(c.get("chip_starks"), dict):
        c["chip_starks"] = {k: (slim(v, keep["chip"]) if isinstance(v, dict) else v) for k, v in c["chip_starks"].items()}
    if isinstance(c.get("rotate"), dict):
        c["rotate"] = slim(c["rotate"], keep["rotate"])
    alone = c.get("recursion_circuits_alone")
    if isinstance(alone, dict):
        c["recursion_circuits_alone"] = {k: (slim(v, keep["alone"]) if k in ("map", "reduce", "outer") else v) for k, v in alone.items() if k != "gate_rows_map"}
    cb = c.get("cpu_baseline")
    if isinstance(cb, dict) and isinstance(cb.get("sampled"), dict):
        cb["sampled"].pop("stages_s", None)
    order = [("alu_bound_dominant_kernel", "ubench_cycles_per_inst"), ("stage_alg_GBps",), ("value_from_host_witness",),
             ("alu_bound_dominant_kernel", "frac_at_pmc_run_clock"), ("alu_bound_dominant_kernel", "frac_quad_issue_model"),
             ("dag_header_range_512", "jobs_by_worker"), ("dag_header_range_512_with_starks", "jobs_by_worker"),
             ("dag_on_one_pool_over_all_gpus", "jobs_by_worker"), ("dag_on_one_pool_over_all_gpus", "stage_elapsed_ms_per_dag"),
             ("sharded_one_proof", "rank0_stage_ms"), ("cpu_baseline", "sampled"),
             ("recursion_circuits_alone", "map_quotient_by_gate_ms_one_kernel_per_gate"),
             ("dag_header_range_512", "stage_elapsed_ms_per_dag"), ("rank_devices",), ("chip_starks",), ("rotate",)]
    dropped = []

    def drop(path):
        if len(path) == 1:
            return c.pop(path[0], None) is not None
        hit = False
        node = c.get(path[0])
        stack = [node] if isinstance(node, dict) else []
        while stack:                                   # the key at any depth below path[0]
            n = stack.pop()
            if path[1] in n:
                del n[path[1]]
                hit = True
            stack.extend(v for v in n.values() if isinstance(v, dict))
        return hit

    for path in order:
        if len(json.dumps(c)) < limit:
            break
        if drop(path):
            dropped.append("/".join(path))
            c["dropped_for_size"] = len(dropped)      # how many detail tables went (their names would cost what they save; the complete line has them)
    return c


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-full-child":
        _cpu_full_child(int(sys.argv[2]), int(sys.argv[3]))

"""Proof-level MapReduce scheduler: the shape of a REAL header_range_N proof.

`verify_subchain` (/root/reference/circuits/builder/subchain_verification.rs:72-78, 233-289) is a plonky2x MapReduce:
N/8 map proofs (8 headers each), a binary tree of reduce proofs that each verify their two children, and one outer
proof on top (/root/reference/circuits/header_range.rs:31-58) — 64 + 63 + 1 = 128 plonky2 proofs for header_range_512.
Proofs inside a layer are independent; a layer can only start when the layer below is done (its proofs are the
witness of the next one).  This module schedules that DAG over the G GPUs of a node, one process per GPU:

  * layer jobs are dealt round-robin to ranks; every rank proves its jobs on its own GPU;
  * the only exchange is an all-gather of the (tiny) proof digests at each layer barrier — proof-level data
    parallelism, no data-path collective; a parent job's public inputs are derived from its children's digests, so
    the dependency is real: no parent can be proven before its children.

The three circuits (map / reduce / outer) are synthetic stand-ins of configurable size (the real circuits need the
Rust builder and the recursion gate set); what is real is the DAG, the barriers and the prover.
The prover behind `make_prover` is pluggable: the GPU library in production, the oracle in the CPU/gloo tests.
"""
from __future__ import annotations

import hashlib
import time
from dataclasses import dataclass

import numpy as np

P = 0xFFFFFFFF00000001


@dataclass
class DagSpec:
    num_map: int = 64              # header_range_512: 512 / HEADERS_PER_MAP(8)  (circuits/consts.rs:6)
    map_log_n: int = 18
    reduce_log_n: int = 16
    outer_log_n: int = 19
    poseidon_percent: int = 50     # only without `recursion`: the PoseidonGate / ArithmeticGate split of rounds 1-5
    recursion: bool = True         # the circuits carry the recursive verifier's gate set in its declared row mix
                                   # (vectorx_amd/synth.py RECURSIVE_VERIFIER_MIX): a reduce job verifies two proofs in-circuit, a map /
                                   # outer job its STARKs (/root/reference/circuits/builder/subchain_verification.rs:78, 233-289)

    def circuit_kw(self):
        return circuit_shape(self.recursion)

    def layers(self):
        """[(kind, [job ids])...]: map layer, reduce layers (binary tree), outer."""
        assert self.num_map >= 2 and self.num_map & (self.num_map - 1) == 0
        out = [("map", list(range(self.num_map)))]
        width = self.num_map // 2
        while width >= 1:
            out.append(("reduce", list(range(width))))
            width //= 2
        out.append(("outer", [0]))
        return out

    def num_proofs(self):
        return sum(len(j) for _, j in self.layers())

    def log_n(self, kind):
        return {"map": self.map_log_n, "reduce": self.reduce_log_n, "outer": self.outer_log_n}[kind]


def circuit_shape(recursion: bool = True) -> dict:
    """SynthCircuit keywords of the DAG's stand-in circuits: with `recursion` PoseidonGate (Merkle paths, challenger), ArithmeticGate,
    ArithmeticExtension, BaseSum, Exponentiation, RandomAccess, MulExtension, Reducing, ReducingExtension, PoseidonMds,
    CosetInterpolation{4 bits, degree 8} and a lookup table in the declared mix; without, the two-gate stand-in of rounds 1-5"""
    from vectorx_amd.synth import RECURSION_FLAGS, RECURSIVE_VERIFIER_MIX
    return {"flags": RECURSION_FLAGS, "mix": RECURSIVE_VERIFIER_MIX} if recursion else {}


def digest_to_field(d: bytes) -> np.ndarray:
    """32-byte digest -> 4 field elements (public inputs of the parent job)"""
    return np.array([int.from_bytes(d[8 * i:8 * i + 8], "little") % P for i in range(4)], dtype=np.uint64)


def child_inputs(layer_idx: int, job: int, prev_digests: dict, input_seed: bytes = b"") -> np.ndarray:
    """Public inputs of a job: H(request input, job) for map jobs, H(left child digest || right child digest) above."""
    if layer_idx == 0:
        return digest_to_field(hashlib.sha256(b"map" + input_seed + job.to_bytes(4, "little")).digest())
    if len(prev_digests) == 1:      # outer proof: single child (the root reduce proof)
        return digest_to_field(hashlib.sha256(b"outer" + prev_digests[0]).digest())
    return digest_to_field(hashlib.sha256(prev_digests[2 * job] + prev_digests[2 * job + 1]).digest())


def _seed_kw(prover, input_seed, children=()):
    """provers that derive per-job table inputs from the request take `input_seed` — and, if they say so, the records of the job's
    children (`takes_children`: a reduce job hashes ITS children's roots); plain ones (the tests' stand-ins) take neither"""
    kw = {"input_seed": input_seed} if getattr(prover, "takes_input_seed", False) else {}
    if getattr(prover, "takes_children", False):
        kw["children"] = tuple(children)
    return kw


STATEMENT_MAGIC = b"VXST"


def with_statement(statement: bytes) -> bytes:
    """a job's statement as the trailer of its result bytes (so that the job's digest covers it): statement | u32 length | magic"""
    return bytes(statement) + len(statement).to_bytes(4, "little") + STATEMENT_MAGIC


def record_of(result: bytes, prover=None) -> bytes:
    """What a job hands to its parent: SHA-256 of everything it proved — followed, when its prover emits statements
    (`emits_statement`, the trailer of with_statement), by the statement itself: 172 bytes of a map / reduce job's subchain
    (vectorx_amd/header_range.py::Subchain), the 96 output bytes of the outer job.  Plain provers: the 32-byte digest of rounds 1-4."""
    dg = hashlib.sha256(result).digest()
    if getattr(prover, "emits_statement", False) and result[-4:] == STATEMENT_MAGIC:
        n = int.from_bytes(result[-8:-4], "little")
        return dg + result[len(result) - 8 - n:len(result) - 8]
    return dg


def children_of(layers, li, j, records):
    """the records of job (li, j)'s children, left to right (none for a map job, one for the outer job)"""
    if li == 0:
        return []
    prev = records[li - 1] if isinstance(records, list) else records
    return [prev[0]] if len(layers[li - 1][1]) == 1 else [prev[2 * j], prev[2 * j + 1]]


def _run_dag_dependency_driven(layers, provers, in_flight, input_seed, sync):
    """One process, no layer barriers: a job starts as soon as ITS children are proven (a reduce job needs its two children, the
    outer proof the root reduce proof), `in_flight` lanes pull from one ready queue — the upper reduce layers, too narrow to fill
    the lanes on their own, run while map proofs are still being produced.  Same proofs, same digests, same root as the layered
    schedule (public inputs depend on the children only); per-layer times are first-start to last-end and overlap."""
    import queue
    import threading
    n_layers = len(layers)
    digests = [dict() for _ in range(n_layers)]
    proofs = {}
    span = [[None, None] for _ in range(n_layers)]
    ready = queue.Queue()
    lock = threading.Lock()
    errors = []
    total = sum(len(j) for _, j in layers)
    done = [0]

    def children(li, j):
        if li == 0:
            return []
        return [0] if len(layers[li - 1][1]) == 1 else [2 * j, 2 * j + 1]

    def parent(li, j):
        if li + 1 >= n_layers:
            return None
        return (li + 1, 0 if len(layers[li][1]) == 1 else j // 2)

    for j in layers[0][1]:
        ready.put((0, j))

    def lane_main(lane):
        while True:
            item = ready.get()
            if item is None:
                return
            li, j = item
            try:
                kind = layers[li][0]
                with lock:
                    prev = dict(digests[li - 1]) if li else {}
                    if span[li][0] is None:
                        span[li][0] = time.perf_counter()
                proof = provers[kind].prove((li, j), child_inputs(li, j, prev, input_seed), lane,
                                            **_seed_kw(provers[kind], input_seed, children_of(layers, li, j, prev)))
                dg = record_of(proof, provers[kind])
                with lock:
                    digests[li][j] = dg
                    proofs[(li, j)] = proof
                    span[li][1] = time.perf_counter()
                    done[0] += 1
                    finished = done[0] == total
                    par = parent(li, j)
                    if par is not None and all(c in digests[li] for c in children(*par)):
                        ready.put(par)
                if finished:
                    for _ in range(in_flight):
                        ready.put(None)
            except BaseException as e:   # surfaces after the join
                errors.append(e)
                for _ in range(in_flight):
                    ready.put(None)
                return

    t0 = time.perf_counter()
    threads = [threading.Thread(target=lane_main, args=(lane,)) for lane in range(in_flight)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    sync()
    seconds = time.perf_counter() - t0
    per_layer = [{"kind": layers[li][0], "jobs": len(layers[li][1]), "ms": (span[li][1] - span[li][0]) * 1e3} for li in range(n_layers)]
    return {"root": digests[-1][0], "seconds": seconds, "proofs": total, "per_layer": per_layer, "my_proofs": proofs}


def run_dag(spec: DagSpec, make_prover, dist=None, sync=lambda: None, in_flight: int = 1, input_seed: bytes = b"", barriers: bool = True):
    """make_prover(kind, log_n, job_ids) -> object with .prove(job, public_inputs[, lane]) -> proof bytes, prepared
    (circuit loaded, per-job witnesses resident) BEFORE the timed region; witness generation is outside the hot path
    (U9).  in_flight > 1: this rank keeps that many jobs of a layer in flight on its GPU (host threads, one prover
    lane = one context/stream each) — the map/reduce proofs are small enough (2^16..2^18 rows) that a single proof
    leaves the chip partly idle in its latency-bound stages.
    barriers=False (one process only): no layer barriers — a job starts when its own children are done (_run_dag_dependency_driven).
    Returns dict(root=<digest of the outer proof>, seconds=<timed DAG wall time>, proofs=<count>, per_layer=[...])."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    layers = spec.layers()
    # ---- setup (untimed): one prover per circuit kind, holding this rank's jobs of every layer of that kind ----
    my_jobs = []
    for li, (kind, jobs) in enumerate(layers):
        my_jobs.append([j for j in jobs if j % world == rank])
    provers = {}
    for kind in ("map", "reduce", "outer"):
        wanted = [(li, j) for li, (k, _) in enumerate(layers) if k == kind for j in my_jobs[li]]
        if wanted:
            provers[kind] = make_prover(kind, spec.log_n(kind), wanted)
    sync()
    if dist is not None:
        dist.barrier()
    if not barriers:
        if dist is not None and world > 1:
            raise ValueError("the dependency-driven schedule is single-process; ranks exchange digests at layer barriers")
        return _run_dag_dependency_driven(layers, provers, max(1, in_flight), input_seed, sync)
    # ---- timed: layer by layer ----
    t0 = time.perf_counter()
    prev = {}
    per_layer = []
    all_proofs = {}
    for li, (kind, jobs) in enumerate(layers):
        tl = time.perf_counter()
        mine = {}
        if in_flight > 1 and len(my_jobs[li]) > 1:
            import queue
            import threading
            todo = queue.Queue()
            for j in my_jobs[li]:
                todo.put(j)
            errors = []

            def lane_main(lane):
                while True:
                    try:
                        j = todo.get_nowait()
                    except queue.Empty:
                        return
                    try:
                        proof = provers[kind].prove((li, j), child_inputs(li, j, prev, input_seed), lane,
                                                    **_seed_kw(provers[kind], input_seed, children_of(layers, li, j, prev)))
                        mine[j] = record_of(proof, provers[kind])
                        all_proofs[(li, j)] = proof
                    except BaseException as e:   # surfaces after the join
                        errors.append(e)
                        return

            threads = [threading.Thread(target=lane_main, args=(lane,)) for lane in range(min(in_flight, len(my_jobs[li])))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
        else:
            for j in my_jobs[li]:
                pi = child_inputs(li, j, prev, input_seed)
                proof = provers[kind].prove((li, j), pi, **_seed_kw(provers[kind], input_seed, children_of(layers, li, j, prev)))
                mine[j] = record_of(proof, provers[kind])
                all_proofs[(li, j)] = proof
        sync()
        if dist is not None:          # layer barrier: all-gather of the digests (32 B per proof)
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            prev = {}
            for g in gathered:
                prev.update(g)
        else:
            prev = mine
        assert sorted(prev) == jobs, "a layer's jobs were not all proven"
        per_layer.append({"kind": kind, "jobs": len(jobs), "ms": (time.perf_counter() - tl) * 1e3})
    seconds = time.perf_counter() - t0
    return {"root": prev[0], "seconds": seconds, "proofs": spec.num_proofs(), "per_layer": per_layer, "my_proofs": all_proofs}


def prove_with_tables(prove_main, tables, lane_ctx=None, split=None, lock=None, job=None, spent_out=None, main_label="plonky2") -> bytes:
    """One job of the DAG = its plonky2 proof followed by the proofs of the STARK tables its circuit embeds (`tables`: [(label, object
    with .prove(ctx, job) -> bytes)]): the concatenation is what the job's digest — and so its parent's public inputs — covers.
    `job` = (kind, layer, index, input_seed): what a per-job table derives ITS inputs from (the headers a map job hashes, the keys and
    signatures of the outer job); a table with one resident trace ignores it.  A table may report finer-grained work through
    `table.last_spent(ctx)` -> [(label, seconds)] (trace generation next to proving).
    `split` (dict, guarded by `lock`) accumulates the wall seconds spent per kind of work; `spent_out` (list) receives this job's own."""
    t0 = time.perf_counter()
    parts = [prove_main()]
    spent = [(main_label, time.perf_counter() - t0)] if main_label else []
    for label, table in tables:
        t0 = time.perf_counter()
        parts.append(table.prove(lane_ctx, job))
        dt = time.perf_counter() - t0
        fine = table.take_spent(lane_ctx) if hasattr(table, "take_spent") else None
        if fine:
            spent.extend(fine)
            dt -= sum(x for _, x in fine)
        spent.append((label, dt))
    if spent_out is not None:
        spent_out.extend(spent)
    if split is not None:
        if lock is not None:
            lock.acquire()
        try:
            for label, dt in spent:
                split[label] = split.get(label, 0.0) + dt
        finally:
            if lock is not None:
                lock.release()
    return b"".join(parts)


def prove_rotate(prover, input_seed: bytes = b"", spent_out=None, ahead=None) -> dict:
    """ONE rotate request (/root/reference/circuits/rotate.rs:80-109 is a single proof, no MapReduce) on `prover` — a GpuProver of kind
    "rotate" carrying dag_tables.build_rotate's tables: the plonky2 proof, the tables of the request `input_seed` names, the statement.
    ahead: dag_tables.AheadTable (build_rotate(bus_lane=...)'s rec["ahead"]) — the signature bus starts NOW on its own lane and the
    job waits for it where the bus stands in its order; same bytes, the wall time of the longer of the two.
    -> {"record": digest || new authority set hash, "output": the 32 output bytes, "seconds"}"""
    t0 = time.perf_counter()
    pis = digest_to_field(hashlib.sha256(b"rotate" + bytes(input_seed)).digest())
    if ahead is not None:
        ahead.start(("rotate", 0, 0, input_seed, ()))
    result = prover.prove((0, 0), pis, 0, input_seed=input_seed, spent_out=spent_out)
    rec = record_of(result, prover)
    return {"record": rec, "output": rec[32:], "seconds": time.perf_counter() - t0, "result_bytes": len(result)}


class GpuProver:
    """One circuit kind on one GPU: circuit loaded once (constants_sigmas resident), one device-resident witness per
    job; `prove` patches the two witness rows that depend on the public inputs and calls vx_prove."""
    takes_input_seed = True
    takes_children = True

    def __init__(self, ctx, kind, log_n, jobs, poseidon_percent=50, extra_lanes=(), distinct_witnesses=None, starks=(), split=None,
                 recursion=True):
        """extra_lanes: more contexts on the SAME GPU; lane k proves on its own stream with its own copy of the circuit
        (run_dag(in_flight=...)).  distinct_witnesses = None: every job has its own generated witness (device-global, shared by
        the lanes).  distinct_witnesses = k: only k base witnesses are generated per circuit kind and job j proves base witness
        j mod k with ITS public inputs patched in — every lane then holds its own copies (two jobs in flight must never patch
        the same buffer).  Same proving work per job, a fraction of the (untimed, CPU) witness generation: the bench's DAG leg.
        starks: [(label, table)] — the STARK tables every job of this kind proves NEXT TO its plonky2 proof (`table.prove(ctx)` -> bytes:
        vectorx_amd.stark_chips.ResidentTable): in the reference every map / reduce / outer circuit embeds Curta STARKs
        (/root/reference/circuits/builder/header.rs:18, justification.rs:140-156, 237-243); a job's result is the plonky2 proof followed
        by its STARK proofs, so the digest the parent consumes covers them.  split: dict label -> lane-seconds, accumulated here."""
        import threading

        import vectorx_amd as vx
        from vectorx_amd.synth import SynthCircuit
        self.ctx, self.n, self.kind = ctx, 1 << log_n, kind
        self.lanes = [ctx] + list(extra_lanes)
        self._lock = threading.Lock()
        self.starks, self.split = list(starks), split
        # a table that `needs_children` is the job's STATEMENT (dag_tables.JobStatement): host logic over what the other tables hashed
        # and what the children stated; it closes the job's result bytes
        self.emits_statement = any(getattr(t, "needs_children", False) for _, t in self.starks)
        circuit_seed = {"map": 101, "reduce": 202, "outer": 303, "rotate": 404}[kind]
        shape = circuit_shape(recursion)
        self.recursion = bool(recursion)
        import os
        trace = os.environ.get("VX_POOL_TRACE")
        t_ = time.perf_counter()
        self.sc = SynthCircuit(log_n, seed=circuit_seed, poseidon_percent=poseidon_percent, witness_seed=0, **shape)
        t1_ = time.perf_counter()
        self.circuits = [vx.Circuit(c, self.sc.desc_ptr) for c in self.lanes]
        if trace:
            import sys
            print(f"[vx pool {os.getpid()}] {kind}: circuit generated in {t1_ - t_:.2f} s, loaded on {len(self.lanes)} lanes in {time.perf_counter() - t1_:.2f} s",
                  file=sys.stderr, flush=True)
        self.circuit = self.circuits[0]
        self.wit = {}
        self.distinct = distinct_witnesses
        self.lane_wit = []
        if distinct_witnesses is None:
            for (li, j) in jobs:
                sj = SynthCircuit(log_n, seed=circuit_seed, poseidon_percent=poseidon_percent, witness_seed=1000 * li + j + 1, **shape)
                w = sj.witness()
                d = ctx.alloc(w.nbytes)
                ctx.upload(d, w)
                self.wit[(li, j)] = d
                sj.free()
        else:
            k = max(1, min(distinct_witnesses, len(jobs)))
            self.distinct = k
            self.lane_wit = [[] for _ in self.lanes]
            for b in range(k):
                sj = SynthCircuit(log_n, seed=circuit_seed, poseidon_percent=poseidon_percent, witness_seed=7000 + b, **shape)
                w = sj.witness()
                for li_, c in enumerate(self.lanes):
                    d = ctx.alloc(w.nbytes)
                    ctx.upload(d, w)
                    self.lane_wit[li_].append(d)
                sj.free()
        self.sc.release_host_buffers(witness=True, preprocessed=True)
        if trace:
            print(f"[vx pool {os.getpid()}] {kind}: witnesses ready {time.perf_counter() - t_:.2f} s after the start", file=sys.stderr, flush=True)

    def prove(self, key, public_inputs, lane=0, input_seed=b"", spent_out=None, with_tables=True, children=()):
        d = self.wit[key] if self.distinct is None else self.lane_wit[lane][key[1] % self.distinct]
        with self._lock:                      # the generator keeps the current public inputs: one caller at a time
            r0, r2 = self.sc.patch_public_inputs(public_inputs)
        ctx = self.lanes[lane]
        ctx.upload_row(d, self.n, 0, r0)
        ctx.upload_row(d, self.n, 2, r2)
        return prove_with_tables(lambda: self.circuits[lane].prove(dev_ptr=d), self.starks if with_tables else (), ctx, self.split, self._lock,
                                 job=(self.kind, key[0], key[1], input_seed, tuple(children)), spent_out=spent_out)

    def prove_tables(self, key, lane=0, input_seed=b"", spent_out=None) -> bytes:
        """the STARK proofs of job `key` alone, in the order `prove` appends them: for a job whose tables do not depend on its children
        (the outer job's: vectorx_amd/dag_pool.py proves them while the reduce tree is still running) — without the job's statement,
        which does (prove_statement)"""
        tables = [(l, t) for l, t in self.starks if not getattr(t, "needs_children", False)]
        return prove_with_tables(lambda: b"", tables, self.lanes[lane], self.split, self._lock, job=(self.kind, key[0], key[1], input_seed, ()),
                                 spent_out=spent_out, main_label=None)

    def prove_statement(self, key, lane=0, input_seed=b"", children=(), spent_out=None) -> bytes:
        """the statement of job `key` alone (the trailer `prove` ends with), from what the tables proven LAST on this lane hashed"""
        tables = [(l, t) for l, t in self.starks if getattr(t, "needs_children", False)]
        return prove_with_tables(lambda: b"", tables, self.lanes[lane], self.split, self._lock,
                                 job=(self.kind, key[0], key[1], input_seed, tuple(children)), spent_out=spent_out, main_label=None)

    def free(self):
        for d in self.wit.values():
            self.ctx.free(d)
        for lane in self.lane_wit:
            for d in lane:
                self.ctx.free(d)
        for c in self.circuits:
            c.free()
        self.sc.free()

"""The STARK tables a header_range_512 job mix proves next to its plonky2 proofs — own AIRs standing in for
Curta's chips, sized from the reference's constants.  Two modes:

  * "resident" (rounds 3-4): ONE trace per table kind, generated on the host in numpy before the clock and proven once per job;
  * "per_job"  (round 5): every job derives ITS OWN inputs from the request seed and its position in the DAG — a map job its 8
    header byte strings (/root/reference/circuits/builder/subchain_verification.rs:148-231, builder/header.rs:14-19), every job its
    SHA-256 messages — and the trace is generated ON THE GPU by native kernels (csrc/tracegen.hip.h: vx_trace_*) inside the job's
    time, reported as `trace_generation` lane-seconds.  Tables without a native generator yet keep one resident trace (listed in
    the setup record under "resident_trace").
"""
from __future__ import annotations

import os
import time

import numpy as np


def build_resident(ctx, eddsa_log_n=20, blake_log_n=18, kinds=("map", "reduce", "outer"), small=False):
    """The STARK tables a header_range_512 job mix proves next to its plonky2 proofs — own AIRs standing in for Curta's chips, sized
    from the reference's constants: a MAP job hashes 8 headers of up to MAX_HEADER_SIZE = 35 840 bytes = 280 BLAKE2b blocks each
    (/root/reference/circuits/consts.rs:6-16, builder/header.rs:18) = 2240 compressions = one 2^18-row BLAKE2b table, and two 8-leaf
    SHA-256 trees (subchain_verification.rs:148-231) = 28 compressions = a 2^11-row SHA-256 table; a REDUCE job merges two
    commitments (4 compressions, 2^9 rows); the OUTER proof chains SHA-256 over 300 authority keys (justification.rs:140-156: 600
    compressions, 2^16 rows), hashes 300 signed messages with SHA-512 (117 bytes = 2 blocks each: 2^16 rows) and checks 300 EdDSA
    equations (justification.rs:237-243) = four 2^20-row batched tables of 97 signatures each (ONE resident trace proven four times:
    the proving work does not depend on the values).  -> ({kind: [(label, table)]}, [tables to free], setup record)"""
    from . import blake2b_air, blake2b_bytes_air, eddsa_air, sha256_air, sha512_air, stark_chips
    rec, tables = {}, []
    nopi = np.zeros(0, dtype=np.uint64)
    # small = True (the single-GPU emulation test of the N-rank path): the smallest shapes the tables allow, same code path
    nkeys, nhdr_blocks = (8, 4) if small else (300, 280)
    if small:
        eddsa_log_n = 17

    def resident(label, stark, trace, pis):
        t0 = time.perf_counter()
        tab = stark_chips.ResidentTable(ctx, stark, trace, pis, label)
        tab.prove()                      # warm-up: loads the compiled evaluator, computes and uploads the second-round columns
        tab.drop_host_trace()
        tables.append(tab)
        rec[label]["first_proof_incl_second_round_columns_s"] = round(time.perf_counter() - t0, 2)
        return tab

    def hash_table(label, air, log_n, msgs):
        t0 = time.perf_counter()
        trace, pis, digests = air.generate_trace(log_n, msgs)
        assert len(digests) == len(msgs), f"{label}: {len(digests)} of {len(msgs)} messages fit 2^{log_n} rows"
        stark = air.make_stark(log_n)
        rec[label] = {"rows_log2": log_n, "columns": f"{stark.desc.num_columns} + {stark.desc.num_aux_columns}", "messages": len(msgs),
                      "trace_generation_s": round(time.perf_counter() - t0, 2)}
        return resident(label, stark, trace, pis)

    per_kind = {}
    if "map" in kinds:
        # round 4: the byte / XOR-lookup table (775 + 238 columns, 28 rows per compression: 2240 compressions fit 2^16 rows); round 3's bit
        # table (1063 + 12 columns, 106 rows per compression, 2^18 rows) with VX_DAG_BLAKE2B_BITS=1, for comparison
        import os
        if os.environ.get("VX_DAG_BLAKE2B_BITS"):
            blake = hash_table("blake2b_map", blake2b_air, blake_log_n, [bytes([17 * i & 255]) * (128 * nhdr_blocks) for i in range(8)])
        else:
            blake = hash_table("blake2b_map", blake2b_bytes_air, 16, [bytes([17 * i & 255]) * (128 * nhdr_blocks) for i in range(8)])
        sha_map = hash_table("sha256_map", sha256_air, 11, [bytes([i]) * 64 for i in range(14)])
        per_kind["map"] = [("blake2b", blake), ("sha256", sha_map)]
    if "reduce" in kinds:
        sha_red = hash_table("sha256_reduce", sha256_air, 9, [bytes([i + 50]) * 64 for i in range(2)])
        per_kind["reduce"] = [("sha256", sha_red)]
    if "outer" in kinds or "outer_eddsa" in kinds:
        if "outer" in kinds:
            sha_out = hash_table("sha256_outer", sha256_air, 11 if small else 16, [bytes([i & 255, i >> 8]) * 32 for i in range(nkeys)])
            s512 = hash_table("sha512_outer", sha512_air, 11 if small else 16, [bytes([i & 255, i >> 8]) * 58 + b"x" for i in range(nkeys)])
        t0 = time.perf_counter()
        lay = eddsa_air.Layout()
        cap = eddsa_air.capacity(lay, eddsa_log_n)
        sigs, rs = stark_chips.eddsa_signatures(cap, 8 if not small else 2)
        trace, res = eddsa_air.generate_trace(lay, eddsa_log_n, sigs)
        assert res == rs
        stark = eddsa_air.make_stark(lay, eddsa_log_n)
        ntab = -(-nkeys // cap)
        rec["eddsa_outer"] = {"rows_log2": eddsa_log_n, "columns": f"{stark.desc.num_columns} + {stark.desc.num_aux_columns}", "signatures_per_table": cap,
                              "tables": ntab, "trace_generation_s": round(time.perf_counter() - t0, 2)}
        ed = resident("eddsa_outer", stark, trace, nopi)
        del trace
        if "outer" in kinds:
            per_kind["outer"] = [("sha256", sha_out), ("sha512", s512), ("eddsa", stark_chips.Repeated(ed, ntab))]
        else:
            per_kind["outer_eddsa"] = [("eddsa", stark_chips.Repeated(ed, ntab))]
    return per_kind, tables, rec




# ---- per-job tables (round 5) ---------------------------------------------------------------------------------------------------------
MAX_HEADER_BLOCKS = 280       # MAX_HEADER_SIZE = 35 840 bytes (/root/reference/circuits/consts.rs:16) = 280 BLAKE2b blocks
HEADERS_PER_MAP = 8           # /root/reference/circuits/consts.rs:6


def job_bytes(job, label: bytes, count: int, length: int) -> list:
    """`count` byte strings of `length` bytes that belong to THIS job and nobody else: SHAKE-256 of (request seed, job kind, layer,
    index, label).  job = (kind, layer, index, input_seed, ...) as mapreduce.prove_with_tables hands it on; None = a fixed job."""
    import hashlib
    kind, li, j, seed = job[:4] if job is not None else ("none", 0, 0, b"")
    h = hashlib.shake_256(b"vectorx job bytes|" + bytes(seed) + b"|" + kind.encode() + b"|%d|%d|" % (li, j) + label)
    blob = h.digest(count * length)
    return [blob[i * length:(i + 1) * length] for i in range(count)]


def request_shape(small: bool, num_map: int, num_headers=None) -> dict:
    """the synthetic request the per-job tables hash (vectorx_amd/header_range.py::make_request): header_range_512's capacity, headers
    of MAX_HEADER_SIZE bytes and 300 authorities — or the miniature of the tests"""
    shape = {"capacity": HEADERS_PER_MAP * num_map, "header_bytes": 128 * (4 if small else MAX_HEADER_BLOCKS), "num_authorities": 8 if small else 300,
             "distinct_keys": 2 if small else 8}
    if num_headers is not None:
        shape["num_headers"] = num_headers
    return shape


def preload_request(seed: bytes, shape: dict, outer: bool):
    """Derive (and keep) the request of `seed` in this process BEFORE the clock: the header chain — and, where the outer job runs, the
    justification with its signatures and the witnesses of the signature equations.  The reference fetches the same from the Avail RPC
    before it proves (/root/reference/circuits/input/mod.rs).  A request that was not preloaded is derived inside the first job."""
    from . import header_range as hr
    req = hr.cached_request(seed, **shape)
    if outer:
        _signature_inputs(req)
    return req


def _signature_inputs(req):
    """-> (raw [(pk, msg, sig)], full-program witnesses) of the validators marked as signed, kept on the request"""
    from . import eddsa_air
    just = req.justification()
    with req._lock:
        if getattr(req, "_sig_inputs", None) is None:
            eq_by_key = {}
            raw, eq = [], []
            for pk, sg, signed in zip(just.pubkeys, just.signatures, just.validator_signed):
                if not signed:
                    continue
                if pk not in eq_by_key:
                    eq_by_key[pk] = eddsa_air.equation_inputs_full(pk, just.encoded_precommit, sg)
                raw.append((pk, just.encoded_precommit, sg))
                eq.append(eq_by_key[pk])
            req._sig_inputs = (raw, eq)
        return req._sig_inputs


class JobStatement:
    """The last "table" of a job: no proof, the job's STATEMENT — host logic (vectorx_amd/header_range.py, the mirror of the circuit's
    assertions) over what the job's tables hashed on this lane and what its children stated — as the trailer of the job's result
    bytes (mapreduce.with_statement), from where the scheduler hands it to the parent (mapreduce.record_of)."""
    needs_children = True

    def __init__(self, fn):
        self.fn = fn

    def prove(self, ctx=None, job=None) -> bytes:
        from . import mapreduce as mr
        return mr.with_statement(self.fn(ctx, job))

    def free(self):
        pass


class GpuTables:
    """what build_per_job makes its tables with: the GPU library's (stark_chips.GeneratedHashTable / GeneratedSignatureBus).  The
    CPU tests hand in a factory of their own with hashlib in the tables' place (tests/_cpu_tables.py): the statements, the records and
    the schedulers above them then run — also over gloo ranks — without a GPU."""

    def __init__(self, ctx):
        self.ctx = ctx

    def hash_table(self, label, which, log_n, messages_fn, lanes):
        from . import blake2b_bytes_air, sha256_air, stark_chips
        air = {"blake2b": blake2b_bytes_air, "sha256": sha256_air}[which]
        # starky's transcript order (the opening set absorbed element by element) unless VX_OPENINGS_DIGEST=1 asks for the tree-hash variant
        stark = air.make_stark(log_n, openings_digest=stark_chips.openings_digest_default())
        tab = stark_chips.GeneratedHashTable(self.ctx, which, stark, log_n, messages_fn, lanes, label)
        return tab, f"{stark.desc.num_columns} + {stark.desc.num_aux_columns}"

    def signature_bus(self, sigs_fn, lanes, nsigs, sha_log_n, ed_log_n):
        from . import stark_chips
        return stark_chips.GeneratedSignatureBus(self.ctx, sigs_fn, lanes, nsigs, sha_log_n=sha_log_n, ed_log_n=ed_log_n)


def build_per_job(ctx, lanes, kinds=("map", "reduce", "outer"), small=False, eddsa_log_n=20, outer_lanes=None, num_map=64, num_headers=None, factory=None):
    """Every job proves tables of ITS OWN inputs — the request's (header_range.make_request(input_seed)) and its children's
    statements —, traces generated on the GPU inside the job (vx_trace_*), and states what the reference's circuit would
    (/root/reference/circuits/builder/subchain_verification.rs:84-289, justification.rs:195-257), with every hash taken from the tables:
      map    : BLAKE2b over its 8 headers (280 blocks each: 2240 compressions, 2^16 rows) + SHA-256 over the 14 nodes of the two
               8-leaf trees of their state / data roots (2^11 rows) -> Subchain (linked headers, block numbers, the two roots);
      reduce : SHA-256 over left root || right root of both trees (2^9 rows) -> the merged Subchain (the right one continues the left);
      outer  : SHA-256 over the authority set commitment chain (300 keys: 599 compressions, 2^16 rows) + the justification's 300
               signatures over the precommit message, verified through tables only as ONE bus: SHA-512 over R || A || M (2^16
               rows), 4 batched EdDSA tables running the full program (three of 2^20 rows with 97 instances each, the last of 2^17 rows for the 9 left over), the link table, the
               verifier's sink -> the 96 output bytes (target header hash, state / data root commitments).
    -> ({kind: [(label, table)]}, [tables to free], setup record)"""
    from . import header_range as hr
    lanes = list(lanes)
    factory = GpuTables(ctx) if factory is None else factory
    rec, tables, per_kind = {"mode": "per_job", "resident_trace": []}, [], {}
    shape = request_shape(small, num_map, num_headers)
    nkeys = shape["num_authorities"]
    rec["request"] = dict(shape, what="synthetic header_range request derived from the seed: headers chained by their BLAKE2b-256 hashes, real Ed25519 "
                                      "authorities, a signed precommit (vectorx_amd/header_range.py::make_request)")
    WARM = b"warm-up"

    def request(job):
        return hr.cached_request(job[3] if job is not None else WARM, **shape)

    def gen(label, which, log_n, messages_fn):
        t0 = time.perf_counter()
        tab, columns = factory.hash_table(label, which, log_n, messages_fn, lanes)
        for lane in lanes:                      # warm-up per lane: loads the compiled evaluator / second-round program, fills the pool
            tab.prove(lane, None)
            tab.take_spent(lane)
        tables.append(tab)
        rec[label] = {"rows_log2": log_n, "columns": columns, "messages": len(messages_fn(None)),
                      "trace": "generated per job on the GPU", "setup_incl_one_proof_per_lane_s": round(time.perf_counter() - t0, 2)}
        return tab

    if "map" in kinds:
        def headers_of(job):
            return request(job).batch(job[2] if job is not None else 0)

        def tree_nodes_of(job):
            req = request(job)
            state, data = hr.map_leaves(req.target_block, headers_of(job))
            return hr.tree_messages(state) + hr.tree_messages(data)

        blake = gen("blake2b_map", "blake2b", 16, headers_of)
        sha_map = gen("sha256_map", "sha256", 11, tree_nodes_of)

        def map_statement(lane, job):
            req = request(job)
            (hdrs, hashes), (tmsgs, tdigs) = blake.last[id(lane)], sha_map.last[id(lane)]
            return hr.map_statement(req.trusted_block, req.target_block, job[2], hdrs, hashes, tmsgs, tdigs).pack()

        per_kind["map"] = [("blake2b", blake), ("sha256", sha_map), ("statement", JobStatement(map_statement))]
    if "reduce" in kinds:
        def children_of(job):
            if job is None or len(job) < 5 or len(job[4]) != 2:
                raise hr.StatementError("a reduce job needs the records of its two children")
            return [hr.Subchain.unpack(bytes(r)[32:]) for r in job[4]]

        def merge_nodes_of(job):
            if job is None:
                return job_bytes(None, b"merge", 2, 64)
            return hr.reduce_messages(*children_of(job))

        sha_red = gen("sha256_reduce", "sha256", 9, merge_nodes_of)

        def reduce_statement(lane, job):
            msgs, digs = sha_red.last[id(lane)]
            return hr.reduce_statement(*children_of(job), msgs, digs).pack()

        per_kind["reduce"] = [("sha256", sha_red), ("statement", JobStatement(reduce_statement))]
    if "outer" in kinds:
        lg = 11 if small else 16
        if outer_lanes is not None:          # a scheduler that only ever proves the outer job on one lane need not hold its buffers on all
            lanes = list(outer_lanes)
        t0 = time.perf_counter()
        preload_request(WARM, shape, outer=True)
        sha_out = gen("sha256_outer", "sha256", lg, lambda job: hr.authority_chain_messages(request(job).justification().pubkeys))
        # The justification's signatures, verified THROUGH TABLES ONLY and proven as one bus (stark_chips.GeneratedSignatureBus): the SHA-512
        # table over the signed messages R || A || precommit (117 bytes each) sends (R, A, digest); four batched EdDSA tables running the
        # FULL program (decompression, digest mod L, S < L, the group equation) send (A, S, digest, R); the link table joins them and
        # sends what a verifier holds; the sink receives it from the bytes of the keys and signatures alone.  REAL Ed25519 signatures
        # of the request's precommit (RFC 8032 signing on the host, untimed: they are the request's input), by 8 distinct authorities.
        from . import eddsa_air
        lg_ed = 17 if small else eddsa_log_n
        bus = factory.signature_bus(lambda job: _signature_inputs(request(job)), lanes, nkeys, lg, lg_ed)
        for lane in lanes:
            bus.prove(lane, None)
            bus.take_spent(lane)
            raw, results, sums = bus.last[id(lane)]
            assert results == [eddsa_air.decompress(sig[:32]) for _, _, sig in raw], "a generated EdDSA instance does not arrive at R"
            assert bus.closed(lane), "the signature bus does not balance"
        tables.append(bus)
        rec["signature_bus"] = {"tables": bus.describe() if hasattr(bus, "describe") else f"SHA-512 bus variant 2^{lg} x 2012 + {bus.ntab} x EdDSA full program 2^{lg_ed} x {bus.lay.N} (the last one 2^{getattr(bus, 'tail_log_n', lg_ed)}: sized to the signatures left over) + link 2^{bus.link_log_n} x 41 + verifier sink 2^{bus.link_log_n} x 26",
                                "signatures": nkeys, "signatures_per_eddsa_table": bus.cap, "eddsa_tables": bus.ntab,
                                "traces": "SHA-512 and EdDSA generated per job on the GPU; link rows written by the host",
                                "proven_as": "one bus: joint challenges over the 7 trace caps; the closing sums add up to 0",
                                "setup_incl_signing_and_one_bus_per_lane_s": round(time.perf_counter() - t0, 2)}
        rec["eddsa_outer"] = {"tables": bus.ntab, "rows_log2": lg_ed, "last_table_rows_log2": getattr(bus, "tail_log_n", lg_ed), "signatures_per_table": bus.cap}

        def outer_statement(lane, job):
            if job is None or len(job) < 5 or len(job[4]) != 1:
                raise hr.StatementError("the outer job needs the record of the root reduce job")
            req = request(job)
            chain_msgs, chain_digs = sha_out.last[id(lane)]
            raw = bus.last[id(lane)][0]
            if not bus.closed(lane):
                raise hr.StatementError("the signature bus does not balance: a signature of the justification does not verify")
            return hr.outer_statement(req.input_bytes, hr.Subchain.unpack(bytes(job[4][0])[32:]), req.justification(), chain_msgs, chain_digs, raw)

        per_kind["outer"] = [("sha256", sha_out), ("signature_bus", bus), ("statement", JobStatement(outer_statement))]
    return per_kind, tables, rec


def rotate_shape(small: bool) -> dict:
    return {"num_authorities": 8 if small else 300, "distinct_keys": 2 if small else 8, "new_authorities": 8 if small else 300}


class AheadTable:
    """A table proven AHEAD of its place in the job, on a lane of its own: `start(job)` launches its proof on `lane` in a host thread
    (the C calls release the GIL), `prove` — called where the table stands in the job's order — waits for it and returns the same
    bytes as proving it there would.  The rotate request runs its signature bus (0.5 s) this way, next to its plonky2 proof and hash
    tables (0.1 s)."""

    def __init__(self, table, lane):
        self.table, self.lane = table, lane
        self._thread, self._out, self._err = None, None, None

    def start(self, job):
        import threading

        def run():
            try:
                self._out = self.table.prove(self.lane, job)
            except BaseException as e:      # surfaces in prove()
                self._err = e

        self._out, self._err = None, None
        self._thread = threading.Thread(target=run)
        self._thread.start()

    def prove(self, ctx=None, job=None) -> bytes:
        if self._thread is None:            # not started ahead: prove it here, on its own lane
            return self.table.prove(self.lane, job)
        self._thread.join()
        self._thread = None
        if self._err is not None:
            raise self._err
        return self._out

    def take_spent(self, ctx=None):
        return self.table.take_spent(self.lane)

    def free(self):
        pass


def build_rotate(ctx, lanes=None, small=False, eddsa_log_n=20, factory=None, bus_lane=None):
    """The tables of ONE rotate proof (/root/reference/circuits/rotate.rs:80-109, builder/rotate.rs:278-323) over a synthetic request
    (header_range.make_rotate_request(input_seed)): BLAKE2b over the epoch end header (2^16 rows: the XOR table's height), SHA-256 over
    the CURRENT authority set's commitment chain and the NEW set's (2 x 599 compressions: 2^17 rows), the justification's 300
    signatures through the signature bus — traces generated on the GPU inside the job — and the job's statement: the new authority
    set's hash, after the checks of header_range.rotate_statement.  bus_lane: a second context on which the bus is proven AHEAD, next to
    the plonky2 proof and the hash tables (mapreduce.prove_rotate starts it: rec["ahead"]).
    -> ({"rotate": [(label, table)]}, [tables to free], record)"""
    from . import header_range as hr
    lanes = list(lanes or [ctx])
    factory = GpuTables(ctx) if factory is None else factory
    shape = rotate_shape(small)
    rec, tables = {"mode": "per_job", "request": dict(shape, what="synthetic rotate request: an epoch end header announcing the new set, justified by the current one")}, []
    WARM = b"warm-up"

    def request(job):
        return hr.cached_request(job[3] if job is not None else WARM, rotate=True, **shape)

    def gen(label, which, log_n, messages_fn):
        t0 = time.perf_counter()
        tab, columns = factory.hash_table(label, which, log_n, messages_fn, lanes)
        for lane in lanes:
            tab.prove(lane, None)
            tab.take_spent(lane)
        tables.append(tab)
        rec[label] = {"rows_log2": log_n, "columns": columns, "messages": len(messages_fn(None)),
                      "trace": "generated per job on the GPU", "setup_incl_one_proof_per_lane_s": round(time.perf_counter() - t0, 2)}
        return tab

    _signature_inputs(request(None))
    blake = gen("blake2b_header", "blake2b", 16, lambda job: [request(job).header])
    sha = gen("sha256_chains", "sha256", 11 if small else 17,
              lambda job: hr.authority_chain_messages(request(job).justification().pubkeys) + hr.authority_chain_messages(request(job).new_pubkeys))
    t0 = time.perf_counter()
    lg, lg_ed = (11, 17) if small else (16, eddsa_log_n)
    bus_lanes = lanes if bus_lane is None else [bus_lane]
    bus = factory.signature_bus(lambda job: _signature_inputs(request(job)), bus_lanes, shape["num_authorities"], lg, lg_ed)
    for lane in bus_lanes:
        bus.prove(lane, None)
        bus.take_spent(lane)
        assert bus.closed(lane), "the signature bus does not balance"
    tables.append(bus)
    rec["signature_bus"] = {"eddsa_tables": bus.ntab, "last_eddsa_table_rows_log2": getattr(bus, "tail_log_n", lg_ed), "signatures": shape["num_authorities"], "setup_incl_one_bus_per_lane_s": round(time.perf_counter() - t0, 2),
                            "proven": "ahead, on a lane of its own, next to the plonky2 proof and the hash tables" if bus_lane is not None else "in the job's order"}
    ahead = AheadTable(bus, bus_lane) if bus_lane is not None else None
    rec["ahead"] = ahead

    def statement(lane, job):
        req = request(job)
        bl = lane if bus_lane is None else bus_lane
        (hdrs, hashes), (cmsgs, cdigs), raw = blake.last[id(lane)], sha.last[id(lane)], bus.last[id(bl)][0]
        if [bytes(h) for h in hdrs] != [req.header]:
            raise hr.StatementError("the BLAKE2b table hashed something else than the epoch end header")
        if not bus.closed(bl):
            raise hr.StatementError("the signature bus does not balance: a signature of the justification does not verify")
        na = req.justification().num_authorities
        return hr.rotate_statement(req.input_bytes, req.header, hashes[0], req.justification(), cmsgs[:na], cdigs[:na], raw, req.start_position,
                                   req.new_pubkeys, cmsgs[na:], cdigs[na:])

    return {"rotate": [("blake2b", blake), ("sha256", sha), ("signature_bus", ahead if ahead is not None else bus), ("statement", JobStatement(statement))]}, tables, rec


def build(ctx, kinds=("map", "reduce", "outer"), small=False, mode="per_job", lanes=None, outer_lanes=None, num_map=64, num_headers=None, factory=None):
    """mode "per_job" (default) or "resident" (rounds 3-4: one host-generated trace per table kind)"""
    if mode == "resident":
        per_kind, tables, rec = build_resident(ctx, kinds=kinds, small=small)
        rec["mode"] = "resident"
        for lane in (lanes or [])[1:]:          # every lane proves every table once (untimed): its pool then holds the STARK shapes
            for tabs in per_kind.values():
                for _, tab in tabs:
                    getattr(tab, "table", tab).prove(lane)
        return per_kind, tables, rec
    if mode != "per_job":
        raise ValueError(mode)
    return build_per_job(ctx, lanes or [ctx], kinds=kinds, small=small, outer_lanes=outer_lanes, num_map=num_map, num_headers=num_headers, factory=factory)

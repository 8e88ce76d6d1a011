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
    index, label).  job = (kind, layer, index, input_seed) as mapreduce.prove_with_tables hands it on; None = a fixed job."""
    import hashlib
    kind, li, j, seed = job if job is not None else ("none", 0, 0, b"")
    h = hashlib.shake_256(b"vectorx job bytes|" + bytes(seed) + b"|" + kind.encode() + b"|%d|%d|" % (li, j) + label)
    blob = h.digest(count * length)
    return [blob[i * length:(i + 1) * length] for i in range(count)]


def build_per_job(ctx, lanes, kinds=("map", "reduce", "outer"), small=False, eddsa_log_n=20, outer_lanes=None):
    """Every job proves tables of ITS OWN inputs, traces generated on the GPU inside the job (vx_trace_*):
      map    : BLAKE2b over its 8 headers (280 blocks each: 2240 compressions, 2^16 rows) + SHA-256 over its 14 tree nodes (2^11 rows);
      reduce : SHA-256 over the 2 nodes that merge its children's commitments (2^9 rows);
      outer  : SHA-256 over the authority set (300 keys: 600 compressions, 2^16 rows) + the justification's 300 signatures verified
               through tables only, as ONE bus: SHA-512 over R || A || M (2^16 rows), 4 batched EdDSA tables running the full program
               (2^20 rows, 97 instances each), the link table.
    -> ({kind: [(label, table)]}, [tables to free], setup record)"""
    from . import blake2b_bytes_air, sha256_air, sha512_air, stark_chips
    lanes = list(lanes)
    rec, tables, per_kind = {"mode": "per_job", "resident_trace": []}, [], {}
    nkeys, nhdr_blocks = (8, 4) if small else (300, MAX_HEADER_BLOCKS)

    def gen(label, which, air, log_n, messages_fn):
        t0 = time.perf_counter()
        stark = air.make_stark(log_n)
        tab = stark_chips.GeneratedHashTable(ctx, which, stark, log_n, messages_fn, lanes, label)
        for lane in lanes:                      # warm-up per lane: loads the compiled evaluator / second-round program, fills the pool
            tab.prove(lane, None)
            tab.take_spent(lane)
        tables.append(tab)
        rec[label] = {"rows_log2": log_n, "columns": f"{stark.desc.num_columns} + {stark.desc.num_aux_columns}", "messages": len(messages_fn(None)),
                      "trace": "generated per job on the GPU", "setup_incl_one_proof_per_lane_s": round(time.perf_counter() - t0, 2)}
        return tab

    if "map" in kinds:
        blake = gen("blake2b_map", "blake2b", blake2b_bytes_air, 16, lambda job: job_bytes(job, b"headers", HEADERS_PER_MAP, 128 * nhdr_blocks))
        sha_map = gen("sha256_map", "sha256", sha256_air, 11, lambda job: job_bytes(job, b"tree", 14, 64))
        per_kind["map"] = [("blake2b", blake), ("sha256", sha_map)]
    if "reduce" in kinds:
        sha_red = gen("sha256_reduce", "sha256", sha256_air, 9, lambda job: job_bytes(job, b"merge", 2, 64))
        per_kind["reduce"] = [("sha256", sha_red)]
    if "outer" in kinds:
        lg = 11 if small else 16
        if outer_lanes is not None:          # a scheduler that only ever proves the outer job on one lane need not hold its buffers on all
            lanes = list(outer_lanes)
        sha_out = gen("sha256_outer", "sha256", sha256_air, lg, lambda job: job_bytes(job, b"authority set", nkeys, 64))
        # The justification's signatures, verified THROUGH TABLES ONLY and proven as one bus (stark_chips.GeneratedSignatureBus): the SHA-512
        # table over the 300 signed messages R || A || M (117 bytes each) sends (R, A, digest); four batched EdDSA tables running the FULL
        # program (decompression, digest mod L, S < L, the group equation) send (A, S, digest, R); the link table joins them and sends what
        # a verifier holds; the sink receives it from the bytes of the keys and signatures alone.  REAL Ed25519 signatures (RFC 8032 signing on the host, untimed: they are the request's input), 8 distinct
        # ones; a job takes them in an order of its own.
        t0 = time.perf_counter()
        from . import eddsa_air
        lg_ed = 17 if small else eddsa_log_n
        base_raw, base_eq = stark_chips.real_signatures(8 if not small else 2)

        def sigs_of(job):
            k = int.from_bytes(job_bytes(job, b"signatures", 1, 2)[0], "little")
            idx = [(k + i) % len(base_raw) for i in range(nkeys)]
            return [base_raw[i] for i in idx], [base_eq[i] for i in idx]

        bus = stark_chips.GeneratedSignatureBus(ctx, sigs_of, lanes, nkeys, sha_log_n=lg, ed_log_n=lg_ed)
        for lane in lanes:
            bus.prove(lane, None)
            bus.take_spent(lane)
            raw, results, sums = bus.last[id(lane)]
            assert results == [eddsa_air.decompress(sig[:32]) for _, _, sig in raw], "a generated EdDSA instance does not arrive at R"
            assert bus.closed(lane), "the signature bus does not balance"
        tables.append(bus)
        rec["signature_bus"] = {"tables": f"SHA-512 bus variant 2^{lg} x 2012 + {bus.ntab} x EdDSA full program 2^{lg_ed} x {bus.lay.N} + link 2^{bus.link_log_n} x 41 + verifier sink 2^{bus.link_log_n} x 26",
                                "signatures": nkeys, "signatures_per_eddsa_table": bus.cap, "eddsa_tables": bus.ntab,
                                "traces": "SHA-512 and EdDSA generated per job on the GPU; link rows written by the host",
                                "proven_as": "one bus: joint challenges over the 7 trace caps; the closing sums add up to 0, second rounds on the GPU",
                                "setup_incl_signing_and_one_bus_per_lane_s": round(time.perf_counter() - t0, 2)}
        rec["eddsa_outer"] = {"tables": bus.ntab, "rows_log2": lg_ed, "signatures_per_table": bus.cap}
        per_kind["outer"] = [("sha256", sha_out), ("signature_bus", bus)]
    return per_kind, tables, rec


def build(ctx, kinds=("map", "reduce", "outer"), small=False, mode="per_job", lanes=None, outer_lanes=None):
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
    return build_per_job(ctx, lanes or [ctx], kinds=kinds, small=small, outer_lanes=outer_lanes)

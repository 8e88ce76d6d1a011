"""vectorx_amd/dag_pool.py with the real prover at tiny sizes: two worker PROCESSES of two lanes each on the one GPU, per-job STARK
tables generated on the device — the root digest must equal the one-process scheduler's (mapreduce.run_dag) for the same request."""
import hashlib

import pytest

import vectorx_amd as vx
from vectorx_amd import avail_codec as ac
from vectorx_amd import dag_tables
from vectorx_amd import header_range as hr
from vectorx_amd import mapreduce as mr
from vectorx_amd.dag_pool import DagPool

pytestmark = pytest.mark.gpu


def test_pool_root_equals_the_one_process_root_with_per_job_tables(ctx):
    spec = mr.DagSpec(4, 10, 9, 11)
    # started first: the workers only touch the GPU when wait_ready() configures them
    pool = DagPool(spec, devices=(0,), workers_per_device=2, lanes=2, with_starks=True, small_tables=True, table_mode="per_job").start()
    try:
        ready = pool.wait_ready(timeout=1500)
        assert len(ready) == 2 and {r["device"] for r in ready} == {0}
        pool.load_request(b"request 1")                       # request 2 below is derived inside its first jobs instead
        with_tables = pool.run(b"request 1")
        again = pool.run(b"request 1", schedule="layers")
        assert with_tables["outer_tables_hoisted"] and not again["outer_tables_hoisted"]
        other = pool.run(b"request 2")
        plain = pool.run(b"request 1", with_tables=False)
        # one more plonky2-only pass with HIP-event profiling on every lane of every worker (bench.py: `stage_ms_per_dag`)
        pool.profile(True)
        profiled = pool.run(b"request 1", with_tables=False)
        stages = pool.profile_get()
        pool.profile(False)
    finally:
        pool.close()
    assert profiled["root"] == plain["root"]
    assert stages["hash_leaves"] > 0 and stages["quotient_eval"] >= stages["quotient_program_gates_jit"] > 0 and stages["quotient_lookup_terms"] > 0
    assert with_tables["root"] == again["root"] != other["root"]
    assert plain["root"] != with_tables["root"]
    assert with_tables["split"].get("trace_generation", 0) > 0 and "trace_generation" not in plain["split"]
    assert min(with_tables["jobs_by_worker"]) > 0
    # what the DAG STATES: the outer job's statement — the tail of the root record — is the function's 96 output bytes, and they equal
    # the host computation over the same synthetic request (hashlib over the headers, avail_codec's commitments); every map / reduce
    # record carries its subchain
    shape = dag_tables.request_shape(True, spec.num_map)
    for seed, res in ((b"request 1", with_tables), (b"request 2", other)):
        req = hr.cached_request(seed, **shape)
        assert len(res["root"]) == 32 + 96 and res["root"][32:] == hr.expected_output(req)
        top = hr.Subchain.unpack(res["records"][-2][0][32:])
        assert (top.num_blocks, top.start_block, top.end_block) == (32, req.trusted_block + 1, req.target_block)
        assert top.start_parent == ac.unpack_header_range_input(req.input_bytes)["trusted_header"]
        first = hr.Subchain.unpack(res["records"][0][0][32:])
        assert first.num_blocks == 8 and first.end_header_hash == hashlib.blake2b(req.headers[7], digest_size=32).digest()
    assert len(plain["root"]) == 32

    # the same request through ONE process: run_dag with a GpuProver per kind and the same per-job tables
    per_kind, tables, _ = dag_tables.build(ctx, small=True, mode="per_job", lanes=[ctx], num_map=spec.num_map)
    provers = {}

    def make(kind, log_n, jobs):
        provers[kind] = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, recursion=spec.recursion, distinct_witnesses=4, starks=per_kind[kind])
        return provers[kind]

    try:
        one = mr.run_dag(spec, make, None, ctx.sync, in_flight=1, input_seed=b"request 1")
    finally:
        for p in provers.values():
            p.free()
        for t in tables:
            t.free()
    assert one["root"] == with_tables["root"]


def test_a_range_shorter_than_the_capacity_and_a_broken_chain(ctx):
    """19 of 32 headers: the third map job is enabled up to the target block, the fourth not at all (empty headers, zero leaves, an
    inactive right subchain in the reduce above it) — the output still equals the host computation.  And a request whose header chain
    is broken (one parent hash changed after the fact) is refused by the map job that sees it: StatementError, where the reference's
    witness generation would fail its link check."""
    spec = mr.DagSpec(4, 10, 9, 11)
    shape = dag_tables.request_shape(True, spec.num_map, 19)
    per_kind, tables, _ = dag_tables.build(ctx, small=True, mode="per_job", lanes=[ctx], num_map=spec.num_map, num_headers=19)
    provers = {}

    def make(kind, log_n, jobs):
        provers[kind] = mr.GpuProver(ctx, kind, log_n, jobs, spec.poseidon_percent, recursion=spec.recursion, distinct_witnesses=2, starks=per_kind[kind])
        return provers[kind]

    try:
        res = mr.run_dag(spec, make, None, ctx.sync, in_flight=1, input_seed=b"short range")
        req = hr.cached_request(b"short range", **shape)
        assert req.target_block - req.trusted_block == 19 and req.headers[19:] == [b""] * 13
        assert res["root"][32:] == hr.expected_output(req)
        # break the chain inside batch 1 of another request and prove that batch alone
        bad = hr.cached_request(b"broken", **shape)
        h = bytearray(bad.headers[10])
        h[5] ^= 1
        bad.headers[10] = bytes(h)
        with pytest.raises(hr.StatementError, match="not linked"):
            provers["map"].prove((0, 1), mr.child_inputs(0, 1, {}, b"broken"), 0, input_seed=b"broken")
    finally:
        for p in provers.values():
            p.free()
        for t in tables:
            t.free()


def test_rotate_states_the_new_authority_set_hash(ctx):
    """One rotate request at miniature sizes: plonky2 proof + BLAKE2b over the epoch end header + SHA-256 over both commitment chains +
    the signature bus -> the statement is avail_codec.rotate_output of the same request; a request whose header announces another key
    than the witness's new set is refused."""
    per_kind, tables, _ = dag_tables.build_rotate(ctx, [ctx], small=True)
    prover = mr.GpuProver(ctx, "rotate", 11, [(0, 0)], 50, distinct_witnesses=1, starks=per_kind["rotate"])
    shape = dag_tables.rotate_shape(True)
    try:
        for seed in (b"rotate 1", b"rotate 2"):
            res = mr.prove_rotate(prover, seed)
            req = hr.cached_request(seed, rotate=True, **shape)
            assert len(res["record"]) == 64 and res["output"] == hr.expected_rotate_output(req) == ac.authority_set_commitment(req.new_pubkeys)
        assert mr.prove_rotate(prover, b"rotate 1")["record"] != res["record"]
        # the bus proven AHEAD on a lane of its own: the same record as in the job's order
        lane2 = vx.Context(0)
        pk2, tables2, rec2 = dag_tables.build_rotate(ctx, [ctx], small=True, bus_lane=lane2)
        prover2 = mr.GpuProver(ctx, "rotate", 11, [(0, 0)], 50, distinct_witnesses=1, starks=pk2["rotate"])
        try:
            assert mr.prove_rotate(prover2, b"rotate 2", ahead=rec2["ahead"])["record"] == res["record"]
            assert mr.prove_rotate(prover2, b"rotate 2")["record"] == res["record"]          # not started ahead: proven in place, on its lane
        finally:
            prover2.free()
            for t in tables2:
                t.free()
            lane2.close()
        bad = hr.cached_request(b"rotate 3", rotate=True, **shape)
        bad.new_pubkeys[3] = bytes(32)
        with pytest.raises(hr.StatementError, match="does not announce"):
            mr.prove_rotate(prover, b"rotate 3")
    finally:
        prover.free()
        for t in tables:
            t.free()


def test_signature_bus_balances_only_when_every_signature_verifies(ctx):
    """The outer job's bus (SHA-512 bus variant + EdDSA full program + link + verifier sink, traces generated on the device): every proof
    verifies under the joint challenges and the closing sums add up to zero; with one signature's S replaced by S + 1 the EdDSA table
    arrives at another R, every table still proves, and the bus no longer balances — the verifier refuses the set."""
    import vectorx_amd as vx
    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_chips
    raw, eq = stark_chips.real_signatures(2)
    state = {"raw": raw, "eq": eq}
    bus = stark_chips.GeneratedSignatureBus(ctx, lambda job: (state["raw"], state["eq"]), [ctx], 2, sha_log_n=11, ed_log_n=17)
    try:
        blob = bus.prove(ctx, None)
        descs, proofs = bus.last_bus[id(ctx)]
        assert blob == b"".join(proofs) and len(proofs) == 4
        sums = vx.stark_verify_bus(descs, proofs)
        assert bus.closed(ctx) and sum(int(x) for x in sums[:, 0]) % ea.P == 0 and all(int(x) != 0 for x in sums[:, 0])
        assert bus.last[id(ctx)][1] == [ea.decompress(sig[:32]) for _, _, sig in raw]
        assert bus.prove(ctx, None) == blob                                       # deterministic
        pk, msg, sig = raw[1]
        forged = sig[:32] + ((int.from_bytes(sig[32:], "little") + 1) % ea.ELL).to_bytes(32, "little")
        state["raw"] = [raw[0], (pk, msg, forged)]
        state["eq"] = [eq[0], ea.equation_inputs_full(pk, msg, forged, check=False)]
        bus.prove(ctx, None)
        assert bus.last[id(ctx)][1][1] != ea.decompress(sig[:32])
        assert not bus.closed(ctx)
        descs, proofs = bus.last_bus[id(ctx)]
        with pytest.raises(RuntimeError):
            vx.stark_verify_bus(descs, proofs)
    finally:
        bus.free()


def test_signature_bus_sizes_its_last_eddsa_table_to_the_signatures_left_over(ctx):
    """26 signatures over tables of 2^18 rows (24 instances each): one full table and a LAST table of 2^17 rows for the two left over — the
    rotate request's 300 = 3 x 97 + 9 in small.  Same AIR at two sizes on one bus: every proof verifies under the joint challenges, the bus
    closes, every instance arrives at its R, and a forged signature in the small table opens the bus again."""
    import vectorx_amd as vx
    from vectorx_amd import eddsa_air as ea
    from vectorx_amd import stark_chips
    raw, eq = stark_chips.real_signatures(26)
    state = {"raw": raw, "eq": eq}
    bus = stark_chips.GeneratedSignatureBus(ctx, lambda job: (state["raw"], state["eq"]), [ctx], 26, sha_log_n=13, ed_log_n=18)
    try:
        assert (bus.cap, bus.ntab, bus.tail_log_n, bus.table_log_n(0), bus.table_log_n(1)) == (24, 2, 17, 18, 17)
        blob = bus.prove(ctx, None)
        descs, proofs = bus.last_bus[id(ctx)]
        assert blob == b"".join(proofs) and len(proofs) == 5
        assert [st.desc.degree_bits for st, _ in descs[1:3]] == [18, 17]
        sums = vx.stark_verify_bus(descs, proofs)
        assert bus.closed(ctx) and sum(int(x) for x in sums[:, 0]) % ea.P == 0
        assert bus.last[id(ctx)][1] == [ea.decompress(sig[:32]) for _, _, sig in raw]
        pk, msg, sig = raw[25]                                                   # the second instance of the small table
        forged = sig[:32] + ((int.from_bytes(sig[32:], "little") + 1) % ea.ELL).to_bytes(32, "little")
        state["raw"] = raw[:25] + [(pk, msg, forged)]
        state["eq"] = eq[:25] + [ea.equation_inputs_full(pk, msg, forged, check=False)]
        bus.prove(ctx, None)
        assert not bus.closed(ctx)
        descs, proofs = bus.last_bus[id(ctx)]
        with pytest.raises(RuntimeError):
            vx.stark_verify_bus(descs, proofs)
    finally:
        bus.free()

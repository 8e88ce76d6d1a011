"""Several STARK tables on ONE bus — the glue around `vx_stark_begin / vx_stark_set_aux_challenges / vx_stark_finish2` and
`vx_stark_verify_shared` (include/vxprover.h).  Curta (starkyx v1.0.0, /root/reference/Cargo.lock:7232-7249) proves every chip
— SHA-256, BLAKE2b, Ed25519 (/root/reference/circuits/builder/justification.rs:140-156, 237; header.rs:18) — as its own trace and
ties the traces together with lookups and a bus whose challenges are drawn after ALL traces are committed; this module is that
shape on this library's own framing (own protocol, not Curta's transcript).

Prover:   one session per table -> every trace cap -> joint challenges -> each table's second-round columns + closing sums ->
          one proof per table.
Verifier: trace caps out of the proofs -> the same joint challenges -> each proof verified with them -> closing sums returned;
          `bus_balanced` checks that what was sent equals what was received (the sums cancel mod p).
Host code only; the proving happens in the library."""
from __future__ import annotations

import ctypes

import numpy as np

from . import VX_E_PROOF, VxError, _chk, lib, stark_joint_challenges

P = 0xFFFFFFFF00000001


def prove_tables(ctx, tables):
    """tables: [(stark, trace, public_inputs)] in bus order -> (proofs, shared challenges)"""
    n_shared = {st.desc.num_aux_challenges for st, _, _ in tables}
    if len(n_shared) != 1:
        raise ValueError("every table on the bus declares the same number of shared challenges")
    sessions = []
    try:
        for st, tr, pi in tables:
            sessions.append(st.begin(ctx, tr, pi))
        caps = [st.session_trace_cap(sess) for (st, _, _), (sess, _, _) in zip(tables, sessions)]
        shared = stark_joint_challenges(caps, [st.desc.cap_height for st, _, _ in tables], n_shared.pop())
        proofs = []
        for (st, _, _), (sess, _, t) in zip(tables, sessions):
            _chk(lib().vx_stark_set_aux_challenges(sess, shared.ctypes.data))
            aux, api = st.run_aux(t, shared)
            proofs.append(st.finish(sess, aux, api))
        return proofs, shared
    finally:
        for sess, _, _ in sessions:
            lib().vx_stark_session_free(sess)


def verify_tables(tables, proofs):
    """tables: [(stark, public_inputs)] -> the closing sums of every table (raises VxError when a proof is invalid)"""
    caps = [st.proof_trace_cap(p) for (st, _), p in zip(tables, proofs)]
    shared = stark_joint_challenges(caps, [st.desc.cap_height for st, _ in tables], tables[0][0].desc.num_aux_challenges)
    return [st.verify(pi, p, shared) for (st, pi), p in zip(tables, proofs)]


def bus_balanced(closing_sums) -> bool:
    """every closing sum — one per bus and per challenge set — cancels over the tables (all tables announce the same number)"""
    k = {len(c) for c in closing_sums}
    if len(k) != 1:
        return False
    return all(sum(int(c[i]) for c in closing_sums) % P == 0 for i in range(k.pop()))


def verify_bus(tables, proofs):
    """`vx_stark_verify_bus` is the same check behind the C ABI (one call: joint challenges, every proof, the balance)"""
    from . import stark_verify_bus
    sums = stark_verify_bus(tables, proofs)
    assert bus_balanced(list(sums))           # the Python restatement of the same rule, kept as a cross-check
    return list(sums)

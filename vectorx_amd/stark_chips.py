"""Timing harness for the chip-sized AIRs (sha256_air / blake2b_air / ed25519_air): one function that proves a table repeatedly through
`vx_stark_begin` / `vx_stark_finish` and returns the record `tools/stark_bench.py` prints and `bench.py`'s extra leg embeds.
Caller-side code (trace generation and second-round columns are host arithmetic, done once outside the timed loop); no oracle."""
from __future__ import annotations

import ctypes
import os
import threading
import time

import numpy as np


def openings_digest_default() -> bool:
    """Does the transcript of the DAG's / the bench's chip tables take the tree hash of the opening set (`VX_STARK_OPENINGS_DIGEST`, this
    library's own variant: saves 0.2 - 1.1 ms of dependent host permutations per lone wide table, nothing in the saturated DAG) instead of
    the opening set itself in starky's order?  Off unless VX_OPENINGS_DIGEST=1: figures are quoted under the starky-order transcript."""
    return os.environ.get("VX_OPENINGS_DIGEST", "0") not in ("", "0")


CHIPS = ("sha256", "blake2b", "ed25519")


def bench_chip(ctx, which: str, log_n: int, steps: int = 3, warmup: int = 1, host_trace: bool = False, check: bool = False) -> dict:
    """`ctx`: an open vectorx_amd.Context.  Trace (unless host_trace) AND second-round columns resident in HBM; the aux columns are
    computed once per distinct challenge (same trace => same challenge)."""
    import vectorx_amd as vx
    if which == "ed25519":
        from . import ed25519_air as air
        name, block_bytes, naux = "Ed25519 scalar-multiplication", 0, air.Cols.NAUX
        what = "188 bytes of result, quotient and carries per row looked up in a 256-entry table (log-derivative)"
    elif which == "blake2b":
        from . import blake2b_air as air
        name, block_bytes, naux = "BLAKE2b-256", 128, 6
        what = "message bytes range-checked by a log-derivative lookup into a 256-entry table"
    elif which == "sha256":
        from . import sha256_air as air
        name, block_bytes, naux = "SHA-256", 64, 3
        what = "log-derivative range check"
    else:
        raise ValueError(which)
    t_gen = time.perf_counter()
    n = 1 << log_n
    nblocks = n // air.PERIOD
    stark = air.make_stark(log_n)
    if which == "ed25519":
        trace, pis, _ = air.generate_trace(log_n, int.from_bytes(bytes(range(7, 7 + n // 256)), "little"))
    else:
        msgs = [bytes([i & 255]) * (block_bytes * 7 + 20) for i in range(max(1, nblocks // 8))]     # 8-block messages
        trace, pis, _ = air.generate_trace(log_n, msgs)
    t_gen = time.perf_counter() - t_gen
    L = vx.lib()
    vp = ctypes.c_void_p
    d_trace = ctx.alloc(trace.nbytes)
    ctx.upload(d_trace, trace)
    h_trace = None
    if host_trace:
        h_trace = ctx.host_alloc(trace.shape)
        h_trace[:] = trace
    cap = 1 << 25
    out = np.empty(cap, dtype=np.uint8)
    chal = np.zeros(stark.desc.num_aux_challenges, dtype=np.uint64)     # one challenge per set (the second round is repeated per set)
    naux = stark.desc.num_aux_columns
    d_aux = ctx.alloc(naux * n * 8)
    state = {"chal": None}

    def prove():
        sess = vp()
        if h_trace is not None:
            rc = L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(h_trace.ctypes.data), 0, pis.ctypes.data, chal.ctypes.data, ctypes.byref(sess))
        else:
            rc = L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(d_trace), 1, pis.ctypes.data, chal.ctypes.data, ctypes.byref(sess))
        if rc != 0:
            raise RuntimeError(L.vx_last_error().decode())
        try:
            if state["chal"] != tuple(int(c) for c in chal):
                aux = stark.run_aux(trace, chal)[0]
                ctx.upload(d_aux, aux)
                state["chal"] = tuple(int(c) for c in chal)
            nb = ctypes.c_size_t(cap)
            rc = L.vx_stark_finish(sess, vp(d_aux), 1, None, out.ctypes.data, ctypes.byref(nb))
            if rc != 0:
                raise RuntimeError(L.vx_last_error().decode())
            return nb.value
        finally:
            L.vx_stark_session_free(sess)

    try:
        t_first = time.perf_counter()
        prove()                                              # includes loading (or compiling) the program's kernels
        t_first = time.perf_counter() - t_first
        for _ in range(warmup):
            prove()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            nb = prove()
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        stages = {k: round(v["ms"] / steps, 3) for k, v in ctx.prof().items()}
        ctx.prof_enable(False)
        if check:
            stark.verify(pis, out[:nb].tobytes())
    finally:
        ctx.free(d_trace)
        ctx.free(d_aux)
        if h_trace is not None:
            ctx.host_free(h_trace)
    ev = stages.get("air_quotient_eval_jit", stages.get("air_quotient_eval", 0.0))
    hashing = stages.get("hash_leaves", 0.0) + stages.get("merkle_levels", 0.0)
    prog, npush = air.build_program()
    unit = "double-and-add steps" if which == "ed25519" else "compression blocks"
    return {
        "metric": f"vx_stark_begin + vx_stark_finish proofs/sec ({name} AIR at chip density; own AIR, not Curta's)", "value": 1.0 / dt, "unit": "proofs/sec",
        "ms_per_proof": dt * 1e3, f"{which}_blocks_per_s": nblocks / dt,
        "config": {"workload": f"{name} AIR: {air.Cols.N} + {naux} columns x 2^{log_n} rows ({nblocks} {unit} of {air.PERIOD} rows), {npush} constraints "
                               f"of degree <= 3, program {len(prog)} words, {what} in a second commitment round, rate_bits 1, "
                               "cap_height 4, 84 queries, 16 PoW bits, " + ("trace in page-locked HOST memory at the start of every proof, aux columns resident in HBM"
                                                                         if host_trace else "trace + aux columns resident in HBM"),
                   "columns": f"{air.Cols.N} + {naux}", "trace_bytes": int(trace.nbytes), "proof_bytes": int(nb), "evaluator": "compiled" if "air_quotient_eval_jit" in stages else "interpreted"},
        "stage_ms_per_proof": stages, "evaluator_ms": ev, "hashing_ms": hashing, "evaluator_share": round(ev / (dt * 1e3), 4),
        "hashing_share": round(hashing / (dt * 1e3), 4), "first_proof_seconds_incl_jit": round(t_first, 2), "trace_generation_seconds_host": round(t_gen, 2),
        "steps": steps, "warmup": warmup, "n_gpus": 1, "data": "synthetic", "dtype": "u64 (Goldilocks field, integer modular arithmetic)",
        "trace_cells_per_s": air.Cols.N * n / dt}


class Repeated:
    """`times` proofs of the same resident table, one after the other (the four EdDSA tables of an outer proof share one trace in the bench)"""

    def __init__(self, table, times):
        self.table, self.times = table, times

    def prove(self, ctx=None, job=None) -> bytes:
        return b"".join(self.table.prove(ctx, job) for _ in range(self.times))


class ResidentTable:
    """A STARK table whose trace lives in HBM: `prove()` is vx_stark_begin -> second-round columns -> vx_stark_finish2.  Tables with an
    AuxProgram (round 4: every table the DAG proves) compute their second round ON THE GPU on every proof (vx_stark_aux_columns, a buffer
    per lane) — it is part of what the benches time; the others compute it on the HOST with the table's aux_fn the first time a
    challenge vector is seen and reuse it (the same trace always draws the same challenges; VX_HOST_AUX=1 forces this path).  Trace
    generation is the caller's witness generation (disclosed in every record)."""

    def __init__(self, ctx, stark, trace, public_inputs, name="", d_trace=None):
        """`trace`: host array, uploaded here — or None with `d_trace` = a device buffer that already holds the trace (e.g. filled by
        Context.trace_hash_table); the table takes ownership of it"""
        self.ctx, self.stark, self.name = ctx, stark, name
        self.pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        self.n = 1 << stark.desc.degree_bits
        if d_trace is not None:
            self.trace, self.d_trace = None, d_trace
        else:
            self.trace = np.ascontiguousarray(trace, dtype=np.uint64)
            self.d_trace = ctx.alloc(self.trace.nbytes)
            ctx.upload(self.d_trace, self.trace)
        self.naux = stark.desc.num_aux_columns
        self.d_aux = ctx.alloc(max(8, self.naux * self.n * 8))
        self.chal = np.zeros(max(1, stark.desc.num_aux_challenges), dtype=np.uint64)
        self.api = None
        self.seen = None
        self.cap = 1 << 25
        self.aux_seconds_host = 0.0
        self.proofs = 0
        self.lane_aux = {}
        self._lock = __import__("threading").Lock()
        self.gpu_aux = not os.environ.get("VX_HOST_AUX")        # tables that have an AuxProgram compute their second round on the GPU

    def drop_host_trace(self):
        """after the first proof the host copy is only needed to recompute second-round columns for OTHER challenges"""
        self.trace = None

    def prove(self, ctx=None, job=None) -> bytes:
        """`ctx`: another context of the same device (a lane of the DAG scheduler); the buffers are device-global.  `job` is ignored:
        a resident table proves the same trace for every job (GeneratedHashTable is the per-job form)"""
        import vectorx_amd as vx
        L, vp = vx.lib(), ctypes.c_void_p
        c = self.ctx if ctx is None else ctx
        sess = vp()
        chal = np.zeros_like(self.chal)
        rc = L.vx_stark_begin(c._h, ctypes.cast(self.stark.desc_ptr, vp), vp(self.d_trace), 1, self.pis.ctypes.data if self.pis.size else None,
                              chal.ctypes.data, ctypes.byref(sess))
        if rc != 0:
            raise RuntimeError(L.vx_last_error().decode())
        try:
            d_aux, api_now = self.d_aux, None
            if self.naux and self.stark.aux_program is not None and self.gpu_aux:
                # round 4: the second-round columns are computed ON THE GPU on every proof (vx_stark_aux_columns) — part of what is timed.
                # One buffer per lane: the running sums are scanned in place, so two lanes proving the same table must not share it.
                with self._lock:
                    d_aux = self.lane_aux.get(id(c))
                    if d_aux is None:
                        d_aux = self.lane_aux[id(c)] = self.ctx.alloc(max(8, self.naux * self.n * 8))
                api_now = self.stark.run_aux_gpu(c, self.d_trace, chal[:self.stark.desc.num_aux_challenges], d_aux)
            elif self.naux:
                key = tuple(int(x) for x in chal)
                if self.seen != key:
                    if self.trace is None:
                        raise RuntimeError(f"table {self.name}: new challenges but the host trace was dropped")
                    t0 = time.perf_counter()
                    aux, self.api = self.stark.run_aux(self.trace, chal[:self.stark.desc.num_aux_challenges])
                    self.aux_seconds_host += time.perf_counter() - t0
                    c.upload(self.d_aux, aux)
                    c.sync()
                    self.seen = key
            out = np.empty(self.cap, dtype=np.uint8)            # per call: lanes prove the same table concurrently
            nb = ctypes.c_size_t(self.cap)
            api_arr = api_now if api_now is not None else self.api
            api = None if api_arr is None or api_arr.size == 0 else api_arr.ctypes.data
            rc = L.vx_stark_finish2(sess, vp(d_aux), 1, api, None, out.ctypes.data, ctypes.byref(nb))
            if rc != 0:
                raise RuntimeError(L.vx_last_error().decode())
            self.proofs += 1
            return out[:nb.value].tobytes()
        finally:
            L.vx_stark_session_free(sess)

    def free(self):
        self.ctx.free(self.d_trace)
        self.ctx.free(self.d_aux)
        for d in self.lane_aux.values():
            self.ctx.free(d)
        self.lane_aux = {}


def prove_device_trace(ctx, stark, d_trace: int, pis, d_aux: int, cap: int = 1 << 25) -> bytes:
    """vx_stark_begin -> second-round columns ON THE GPU (the table's AuxProgram) -> vx_stark_finish2, for a trace in device memory;
    `d_aux` = [num_aux_columns][n] scratch of the calling lane."""
    import vectorx_amd as vx
    L, vp = vx.lib(), ctypes.c_void_p
    sess = vp()
    chal = np.zeros(max(1, stark.desc.num_aux_challenges), dtype=np.uint64)
    pis = np.ascontiguousarray(pis, dtype=np.uint64)
    rc = L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(d_trace), 1, pis.ctypes.data if pis.size else None, chal.ctypes.data, ctypes.byref(sess))
    if rc != 0:
        raise RuntimeError(L.vx_last_error().decode())
    try:
        api = None
        if stark.desc.num_aux_columns:
            if stark.aux_program is None:
                raise RuntimeError("prove_device_trace: the table has second-round columns but no AuxProgram to compute them on the GPU")
            api = stark.run_aux_gpu(ctx, d_trace, chal[:stark.desc.num_aux_challenges], d_aux)
        out = np.empty(cap, dtype=np.uint8)
        nb = ctypes.c_size_t(cap)
        rc = L.vx_stark_finish2(sess, vp(d_aux), 1, None if api is None or api.size == 0 else api.ctypes.data, None, out.ctypes.data, ctypes.byref(nb))
        if rc != 0:
            raise RuntimeError(L.vx_last_error().decode())
        return out[:nb.value].tobytes()
    finally:
        L.vx_stark_session_free(sess)


def prove_bus_device(ctx, items, cap: int = 1 << 25):
    """Several tables on ONE bus, every trace already in device memory (vectorx_amd/stark_bus.py is the host-trace form): one session per
    table -> every trace cap -> the joint challenges -> each table's second-round columns ON THE GPU (its AuxProgram) with those
    challenges -> one proof per table.  items: [(stark, d_trace, public_inputs, d_aux)] -> (proofs, shared challenges, closing sums)"""
    import vectorx_amd as vx
    L, vp = vx.lib(), ctypes.c_void_p
    n_shared = {st.desc.num_aux_challenges for st, _, _, _ in items}
    if len(n_shared) != 1:
        raise ValueError("every table on the bus declares the same number of shared challenges")
    sessions = []
    try:
        for st, d_trace, pis, _ in items:
            sess = vp()
            pis = np.ascontiguousarray(pis, dtype=np.uint64)
            own = np.zeros(max(1, st.desc.num_aux_challenges), dtype=np.uint64)
            rc = L.vx_stark_begin(ctx._h, ctypes.cast(st.desc_ptr, vp), vp(d_trace), 1, pis.ctypes.data if pis.size else None, own.ctypes.data, ctypes.byref(sess))
            if rc != 0:
                raise RuntimeError(L.vx_last_error().decode())
            sessions.append(sess)
        caps = [st.session_trace_cap(sess) for (st, _, _, _), sess in zip(items, sessions)]
        shared = vx.stark_joint_challenges(caps, [st.desc.cap_height for st, _, _, _ in items], n_shared.pop())
        proofs, sums = [], []
        for k, ((st, d_trace, _, d_aux), sess) in enumerate(zip(items, sessions)):
            vx._chk(L.vx_stark_set_aux_challenges(sess, shared.ctypes.data))
            if st.aux_program is None:
                raise RuntimeError("prove_bus_device: a table without an AuxProgram cannot compute its second round on the GPU")
            try:
                api = st.run_aux_gpu(ctx, d_trace, shared, d_aux)
            except vx.VxError as e:
                raise vx.VxError(e.code, f"bus table {k} of {len(items)} ({st.desc.num_columns} columns, 2^{st.desc.degree_bits} rows): {e}") from None
            out = np.empty(cap, dtype=np.uint8)
            nb = ctypes.c_size_t(cap)
            rc = L.vx_stark_finish2(sess, vp(d_aux), 1, None if api.size == 0 else api.ctypes.data, None, out.ctypes.data, ctypes.byref(nb))
            if rc != 0:
                raise RuntimeError(L.vx_last_error().decode())
            proofs.append(out[:nb.value].tobytes())
            sums.append(api)
        return proofs, shared, sums
    finally:
        for sess in sessions:
            L.vx_stark_session_free(sess)


class GeneratedHashTable:
    """A hash-chip table whose trace is generated PER JOB on the GPU (round 5): `messages_fn(job)` -> the byte strings this job hashes
    (a map job's 8 headers, a job's SHA-256 tree nodes ...), `Context.trace_hash_table` fills the lane's trace buffer with native
    kernels (vx_trace_*), then the table is proven from that buffer.  One trace + one second-round buffer per lane, allocated up
    front.  `take_spent(ctx)` reports the seconds of the last prove that went into trace generation (incl. deriving the messages)."""

    def __init__(self, ctx, which: str, stark, log_n: int, messages_fn, lanes, name=""):
        self.ctx, self.which, self.stark, self.log_n, self.messages_fn, self.name = ctx, which, stark, log_n, messages_fn, name
        ncols = ctx.TRACE_TABLES[which][0]
        assert ncols == stark.desc.num_columns, (which, ncols, stark.desc.num_columns)
        n = 1 << log_n
        self.bufs = {id(c): (ctx.alloc(ncols * n * 8), ctx.alloc(max(8, stark.desc.num_aux_columns * n * 8))) for c in lanes}
        self.spent = {}
        self.last = {}

    def prove(self, ctx=None, job=None) -> bytes:
        c = self.ctx if ctx is None else ctx
        d_trace, d_aux = self.bufs[id(c)]
        t0 = time.perf_counter()
        msgs = self.messages_fn(job)
        pis, digests = c.trace_hash_table(self.which, self.log_n, msgs, d_trace)
        self.spent[id(c)] = [("trace_generation", time.perf_counter() - t0)]
        self.last[id(c)] = (msgs, digests)
        return prove_device_trace(c, self.stark, d_trace, pis, d_aux)

    def take_spent(self, ctx=None):
        return self.spent.pop(id(self.ctx if ctx is None else ctx), None)

    def free(self):
        for a, b in self.bufs.values():
            self.ctx.free(a)
            self.ctx.free(b)
        self.bufs = {}


class GeneratedEddsaTables:
    """The batched EdDSA tables of ONE job, traces generated per job on the GPU (vx_trace_eddsa): `sigs_fn(job)` -> the job's signature
    equations [((ax, ay), S, h)]; they fill ceil(len / capacity) tables of 2^log_n rows, each generated into the lane's trace buffer and
    proven from it, one after the other.  `take_spent` as GeneratedHashTable."""

    def __init__(self, ctx, stark, lay, log_n: int, sigs_fn, lanes, name="eddsa"):
        from . import eddsa_air as ea
        self.ctx, self.stark, self.lay, self.log_n, self.sigs_fn, self.name = ctx, stark, lay, log_n, sigs_fn, name
        assert lay.LB == 16 and lay.N == stark.desc.num_columns
        self.cap = ea.capacity(lay, log_n)
        n = 1 << log_n
        self.bufs = {id(c): (ctx.alloc(lay.N * n * 8), ctx.alloc(max(8, stark.desc.num_aux_columns * n * 8))) for c in lanes}
        self.spent, self.last = {}, {}
        self.nopi = np.zeros(0, dtype=np.uint64)

    def tables_for(self, nsigs: int) -> int:
        return max(1, -(-nsigs // self.cap))

    def prove(self, ctx=None, job=None) -> bytes:
        c = self.ctx if ctx is None else ctx
        d_trace, d_aux = self.bufs[id(c)]
        sigs = self.sigs_fn(job)
        parts, gen, results = [], 0.0, []
        for t in range(self.tables_for(len(sigs))):
            chunk = sigs[t * self.cap:(t + 1) * self.cap]
            t0 = time.perf_counter()
            results += c.trace_eddsa_table(self.log_n, self.lay.NB, chunk, d_trace, full=self.lay.full)
            gen += time.perf_counter() - t0
            parts.append(prove_device_trace(c, self.stark, d_trace, self.nopi, d_aux))
        self.spent[id(c)] = [("trace_generation", gen)]
        self.last[id(c)] = (sigs, results)
        return b"".join(parts)

    def take_spent(self, ctx=None):
        return self.spent.pop(id(self.ctx if ctx is None else ctx), None)

    def free(self):
        for a, b in self.bufs.values():
            self.ctx.free(a)
            self.ctx.free(b)
        self.bufs = {}


class GeneratedSignatureBus:
    """A job's Ed25519 signatures verified THROUGH TABLES ONLY, proven as one bus (round 5): from the raw (public key, message, signature)
    triples `sigs_fn(job)` returns —
      * the SHA-512 table (bus variant) over R || A || M, its trace generated on the GPU (vx_trace_sha512_bus): sends (R, A, digest);
      * ceil(len / capacity) batched EdDSA tables running the FULL program (the last one sized to the signatures left over), traces
        generated on the GPU (vx_trace_eddsa): each signature sends (A, S, digest, R) after decompressing both points, reducing the digest mod L and checking S < L;
      * the link table (40 words per signature, written by the host): joins the two on the digest, sends (A, S, R);
    all committed first, then the joint challenges, every second round on the GPU, one proof per table (prove_bus_device).  What leaves the
    the verifier's sink (25 words per signature, written by the host from the bytes of the public keys and signatures alone): receives
        exactly what the link table sends — the statement, standing where the job's plonky2 circuit would read it;
    all committed first, then the joint challenges, every second round on the GPU, one proof per table (prove_bus_device).  The closing
    sums of the 3 + ceil(len / capacity) tables add up to zero exactly when every signature the sink names went through all of them —
    `last[lane]` keeps them and `closed(lane)` says so.  One set of buffers per lane."""

    def __init__(self, ctx, sigs_fn, lanes, nsigs: int, sha_log_n: int = 16, ed_log_n: int = 20, name="signature_bus"):
        from . import eddsa_air as ea
        from . import sha512_air as s5
        from . import sig_link_air as link
        self.ctx, self.sigs_fn, self.name = ctx, sigs_fn, name
        self.lay = ea.Layout(full=True)
        self.sha_log_n, self.ed_log_n = sha_log_n, ed_log_n
        self.link_log_n = max(4, (nsigs + 1).bit_length())
        self.cap = ea.capacity(self.lay, ed_log_n)
        self.ntab = max(1, -(-nsigs // self.cap))
        # The LAST table holds what the full ones leave over and is sized to it (round 6): 300 signatures = 3 x 97 + 9, and 9 instances
        # fit 2^17 rows — an eighth of the commitment, LDE and hashing of a 2^20-row table whose rows would be 97 % filler.  Same AIR,
        # same program; only degree_bits differs (2^17 is the smallest size the chip is exercised at: tests/test_gpu_stark.py).
        self.tail_log_n = ed_log_n
        rem = max(1, nsigs - (self.ntab - 1) * self.cap)
        while self.tail_log_n > 17 and ea.capacity(self.lay, self.tail_log_n - 1) >= rem:
            self.tail_log_n -= 1
        # starky's transcript order by default since round 6 (the opening set is absorbed element by element, as an in-circuit verifier
        # of the reference would follow it); VX_OPENINGS_DIGEST=1 = this library's own variant, a tree hash of the openings made on the device
        od = openings_digest_default()
        self.sha = s5.make_stark(sha_log_n, bus=True, openings_digest=od)
        self.ed = ea.make_stark(self.lay, ed_log_n, openings_digest=od)
        self.ed_tail = self.ed if self.tail_log_n == ed_log_n else ea.make_stark(self.lay, self.tail_log_n, openings_digest=od)
        self.link = link.make_stark(self.link_log_n, openings_digest=od)
        self.sink = ea.make_sink(self.lay, [], degree_bits=self.link_log_n, ntuple=link.NVERIFIER, openings_digest=od)[0]
        self._link_mod = link
        sizes = [(2012, self.sha.desc.num_aux_columns, sha_log_n)] + [(self.lay.N, self.ed.desc.num_aux_columns, ed_log_n)] * (self.ntab - 1) \
            + [(self.lay.N, self.ed_tail.desc.num_aux_columns, self.tail_log_n)] \
            + [(link.N, self.link.desc.num_aux_columns, self.link_log_n), (link.NVERIFIER + 1, self.sink.desc.num_aux_columns, self.link_log_n)]
        self.bufs = {id(c): [(ctx.alloc(nc * (8 << lg)), ctx.alloc(max(8, na * (8 << lg)))) for nc, na, lg in sizes] for c in lanes}
        self.spent, self.last, self.last_bus = {}, {}, {}
        # The EdDSA traces of one bus are generated side by side: a table's generator first runs ONE lane per signature through the
        # whole ladder (two wavefronts for 97 signatures, 21 of its 35 ms), so the tables go to streams of their own — one helper
        # context per further table and lane, a host thread each (the C call releases the GIL) — and overlap: 4 tables in ~40 ms
        # instead of 140.  Buffers are the lane's; a helper only lends its stream and its scratch.
        import vectorx_amd as vx
        self.helpers = {id(c): [vx.Context(c.device) for _ in range(self.ntab - 1)] for c in lanes}

    def table_log_n(self, t: int) -> int:
        """log2 rows of EdDSA table t: the full tables' size, the last one's own"""
        return self.tail_log_n if t == self.ntab - 1 else self.ed_log_n

    def prove(self, ctx=None, job=None) -> bytes:
        from . import eddsa_air as ea
        c = self.ctx if ctx is None else ctx
        bufs = self.bufs[id(c)]
        t0 = time.perf_counter()
        raw, eq = self.sigs_fn(job)                     # [(pk, msg, sig)], [((ax, ay), S, h, digest)]
        nopi = np.zeros(0, dtype=np.uint64)
        pis, digests = c.trace_hash_table("sha512_bus", self.sha_log_n, [sig[:32] + pk + msg for pk, msg, sig in raw], bufs[0][0])
        items = [(self.sha, bufs[0][0], pis, bufs[0][1])]
        gens = [c] + self.helpers[id(c)]
        per_table = [None] * self.ntab

        failed = []

        def generate(t):
            try:
                per_table[t] = gens[t].trace_eddsa_table(self.table_log_n(t), 256, eq[t * self.cap:(t + 1) * self.cap], bufs[1 + t][0], full=True)
            except BaseException as e:          # surfaces after the join, with the generator's own message
                failed.append(e)

        if self.ntab > 1:
            c.sync()                                    # the SHA-512 trace is on the lane's stream; the helpers' streams know nothing of it
            threads = [threading.Thread(target=generate, args=(t,)) for t in range(1, self.ntab)]
            for th in threads:
                th.start()
            generate(0)
            for th in threads:
                th.join()
        else:
            generate(0)
        if failed:
            raise failed[0]
        results = [r for part in per_table for r in part]     # (the call returns the instances' results: it has synchronised its stream)
        for t in range(self.ntab):
            items.append((self.ed_tail if t == self.ntab - 1 else self.ed, bufs[1 + t][0], nopi, bufs[1 + t][1]))
        rows = [self._link_mod.row_of(pk, sig, dg) for (pk, _, sig), dg in zip(raw, digests)]
        c.upload(bufs[-2][0], self._link_mod.trace_of(rows, self.link_log_n))
        items.append((self.link, bufs[-2][0], nopi, bufs[-2][1]))
        sink_t = np.zeros((self._link_mod.NVERIFIER + 1, 1 << self.link_log_n), dtype=np.uint64)
        sink_t[:-1, :len(raw)] = np.array([self._link_mod.verifier_tuple(pk, sig) for pk, _, sig in raw], dtype=np.uint64).T
        sink_t[-1, :len(raw)] = 1
        c.upload(bufs[-1][0], sink_t)
        items.append((self.sink, bufs[-1][0], nopi, bufs[-1][1]))
        gen = time.perf_counter() - t0
        proofs, _, sums = prove_bus_device(c, items)
        self.spent[id(c)] = [("trace_generation", gen)]
        self.last[id(c)] = (raw, results, sums)
        self.last_bus[id(c)] = ([(st, pi) for st, _, pi, _ in items], proofs)      # what vx.stark_verify_bus takes
        return b"".join(proofs)

    def closed(self, ctx=None) -> bool:
        """the last bus proven on this lane balances: for EVERY challenge set on its own (a table carries one closing sum per set) the
        tables' closing sums add up to 0 mod p — index by index, as vx_stark_verify_bus checks them; adding the sets together would
        accept X + Y = 0 with X != 0 (single-set soundness)"""
        sums = self.last[id(self.ctx if ctx is None else ctx)][2]
        width = len(sums[0])
        if any(len(s_) != width for s_ in sums):
            return False
        return all(sum(int(s_[i]) for s_ in sums) % 0xFFFFFFFF00000001 == 0 for i in range(width))

    def take_spent(self, ctx=None):
        return self.spent.pop(id(self.ctx if ctx is None else ctx), None)

    def free(self):
        for pairs in self.bufs.values():
            for a, b in pairs:
                self.ctx.free(a)
                self.ctx.free(b)
        self.bufs = {}
        for hs in self.helpers.values():
            for h in hs:
                h.close()
        self.helpers = {}


def real_signatures(distinct: int = 8, seed: int = 2024):
    """`distinct` real Ed25519 signatures (fresh keys, RFC 8032 signing on the host) of 53-byte messages — the size of VectorX's precommit
    message, so that R || A || M is 117 bytes — as raw bytes AND as the full program's inputs: -> ([(pk, msg, sig)], [((ax, ay), S, h, digest)])"""
    from . import eddsa_air as ea
    raw, eq = [], []
    for i in range(distinct):
        sk = bytes([(seed + 7 * i + j) & 255 for j in range(32)])
        msg = (b"precommit %d " % i).ljust(53, b".")
        pk, sig = ea.sign(sk, msg)
        raw.append((pk, msg, sig))
        eq.append(ea.equation_inputs_full(pk, msg, sig))
    return raw, eq


def eddsa_signatures(count: int, distinct: int = 8, seed: int = 2024):
    """`count` signature equations (A, S, h) cycling over `distinct` real Ed25519 signatures (fresh keys, RFC 8032 signing on the host)
    -> (sigs, expected R per entry)"""
    from . import eddsa_air as ea
    base = []
    for i in range(distinct):
        sk = bytes([(seed + 7 * i + j) & 255 for j in range(32)])
        msg = b"precommit %d" % i
        pk, sig = ea.sign(sk, msg)
        a, s, h, r = ea.equation_inputs(pk, msg, sig)
        base.append(((a, s, h), r))
    return [base[i % distinct][0] for i in range(count)], [base[i % distinct][1] for i in range(count)]


def eddsa_signatures_full(count: int, distinct: int = 8, seed: int = 2024):
    """the same signatures as the FULL program takes them (eddsa_air.Layout(full=True)): [((ax, ay), S, h, digest)] and the R every
    instance must arrive at — keys, messages and signatures are real (RFC 8032 signing on the host)"""
    from . import eddsa_air as ea
    base = []
    for i in range(distinct):
        sk = bytes([(seed + 7 * i + j) & 255 for j in range(32)])
        msg = b"precommit %d" % i
        pk, sig = ea.sign(sk, msg)
        base.append((ea.equation_inputs_full(pk, msg, sig), ea.decompress(sig[:32])))
    return [base[i % distinct][0] for i in range(count)], [base[i % distinct][1] for i in range(count)]


def bench_eddsa(ctx, log_n: int = 20, steps: int = 3, warmup: int = 1, distinct: int = 8, check: bool = False) -> dict:
    """the batched EdDSA table (eddsa_air.py) at production shape: as many signature equations as 2^log_n rows hold, trace and
    second-round columns resident in HBM"""
    from . import eddsa_air as ea
    lay = ea.Layout()
    t_gen = time.perf_counter()
    cap = ea.capacity(lay, log_n)
    sigs, rs = eddsa_signatures(cap, distinct)
    stark = ea.make_stark(lay, log_n)
    trace, res = ea.generate_trace(lay, log_n, sigs)
    assert res == rs, "a signature equation does not hold"
    t_gen = time.perf_counter() - t_gen
    nopi = np.zeros(0, dtype=np.uint64)
    tab = ResidentTable(ctx, stark, trace, nopi, "eddsa")
    try:
        t_first = time.perf_counter()
        proof = tab.prove()
        t_first = time.perf_counter() - t_first - tab.aux_seconds_host
        for _ in range(warmup):
            tab.prove()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            proof = tab.prove()
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        stages = {k: round(v["ms"] / steps, 3) for k, v in ctx.prof().items()}
        ctx.prof_enable(False)
        if check:
            sums = stark.verify(nopi, proof)
            assert any(int(s) for s in sums)
    finally:
        tab.free()
    ev = stages.get("air_quotient_eval_jit", stages.get("air_quotient_eval", 0.0))
    hashing = stages.get("hash_leaves", 0.0) + stages.get("merkle_levels", 0.0)
    prog, npush = ea.build_program(lay)
    ncols, naux = lay.N, stark.desc.num_aux_columns
    return {"metric": "batched EdDSA table proofs/sec (own AIR, not Curta's)", "value": 1.0 / dt, "unit": "proofs/sec", "ms_per_proof": dt * 1e3,
            "signatures_per_table": cap, "ms_per_signature": dt * 1e3 / cap, "signature_equations_per_s": cap / dt,
            "config": {"workload": f"[S]B - [h]A = R for {cap} Ed25519 signatures ({distinct} distinct, cycled) in one trace: {ncols} + {naux} columns x 2^{log_n} rows, "
                                   f"{lay.L} rows per signature (16 + 42 x 256 + 4), 16-bit limbs looked up in a 65536-entry table, {npush} constraints per challenge "
                                   f"set of degree <= 3, program {len(prog)} words, second round repeated for 2 challenge sets, rate_bits 1, cap_height 4, 84 queries, "
                                   "16 PoW bits; trace + second-round columns resident in HBM",
                       "columns": f"{ncols} + {naux}", "trace_bytes": int(trace.nbytes), "proof_bytes": len(proof),
                       "evaluator": "compiled" if "air_quotient_eval_jit" in stages else "interpreted"},
            "stage_ms_per_proof": stages, "evaluator_ms": ev, "hashing_ms": hashing, "first_proof_seconds_incl_jit": round(t_first, 2),
            "trace_generation_seconds_host": round(t_gen, 2), "second_round_columns_seconds_host": round(tab.aux_seconds_host, 2),
            "steps": steps, "warmup": warmup, "n_gpus": 1, "data": "synthetic", "trace_cells_per_s": (ncols + naux) * (1 << log_n) / dt}


def bench_table(ctx, stark, trace, public_inputs, name: str, steps: int = 3, warmup: int = 1, extra=None, d_trace=None) -> dict:
    """any table through ResidentTable: trace resident in HBM, second-round columns computed on the GPU in every proof (tables with an
    AuxProgram), HIP-event stage times"""
    t0 = time.perf_counter()
    tab = ResidentTable(ctx, stark, trace, public_inputs, name, d_trace=d_trace)
    try:
        proof = tab.prove()
        t_first = time.perf_counter() - t0 - tab.aux_seconds_host
        for _ in range(warmup):
            tab.prove()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            proof = tab.prove()
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        stages = {k: round(v["ms"] / steps, 3) for k, v in ctx.prof().items()}
        ctx.prof_enable(False)
    finally:
        tab.free()
    d = stark.desc
    rec = {"table": name, "ms_per_proof": dt * 1e3, "rows_log2": d.degree_bits, "columns": f"{d.num_columns} + {d.num_aux_columns}",
           "trace_cells_per_s": (d.num_columns + d.num_aux_columns) * (1 << d.degree_bits) / dt, "proof_bytes": len(proof),
           "stage_ms_per_proof": stages, "evaluator": "compiled" if "air_quotient_eval_jit" in stages else "interpreted",
           "first_proof_seconds_incl_upload": round(t_first, 2), "second_round_columns_seconds_host": round(tab.aux_seconds_host, 2),
           "steps": steps, "warmup": warmup}
    rec.update(extra or {})
    return rec


def bench_blake2b_bytes(ctx, log_n: int = 16, steps: int = 3, warmup: int = 1, compressions: int = 2240) -> dict:
    """the byte / XOR-lookup BLAKE2b table (blake2b_bytes_air.py) loaded like a header_range map job: 8 headers of 280 blocks"""
    from . import blake2b_bytes_air as b2
    t0 = time.perf_counter()
    per = compressions // 8
    msgs = [bytes([17 * i & 255]) * (128 * per) for i in range(8)]
    trace, pis, digests = b2.generate_trace(log_n, msgs)
    assert digests == b2.reference_digests(msgs)
    t_gen = time.perf_counter() - t0
    stark = b2.make_stark(log_n)
    rec = bench_table(ctx, stark, trace, pis, "blake2b_bytes", steps, warmup,
                      {"compressions_in_messages": compressions, "compression_slots": (1 << log_n) // b2.PERIOD, "trace_generation_seconds_host": round(t_gen, 2)})
    rec["compressions_per_s"] = compressions / (rec["ms_per_proof"] * 1e-3)
    return rec

"""hashlib in the place of the DAG's GPU tables (dag_tables.build_per_job(factory=...)): the statements, the records and the
schedulers run as in production — each "table" remembers what it hashed and its digests on the lane that proved it, the "signature bus"
verifies every signature with the RFC 8032 host check — without a GPU.  tests/test_header_range_dag.py, tests/_mp_statement_worker.py."""
import hashlib

from vectorx_amd import eddsa_air as ea
from vectorx_amd import mapreduce as mr

HASH = {"sha256": lambda m: hashlib.sha256(m).digest(), "blake2b": lambda m: hashlib.blake2b(m, digest_size=32).digest()}


class HashlibTable:
    def __init__(self, which, messages_fn):
        self.which, self.messages_fn, self.last = which, messages_fn, {}

    def prove(self, ctx=None, job=None) -> bytes:
        msgs = [bytes(m) for m in self.messages_fn(job)]
        digs = [HASH[self.which](m) for m in msgs]
        self.last[id(ctx)] = (msgs, digs)
        return hashlib.sha256(b"table|" + self.which.encode() + b"".join(digs)).digest()

    def take_spent(self, ctx=None):
        return None

    def free(self):
        pass


class HostBus:
    """every (public key, message, signature) through the RFC 8032 host check; `closed` = all of them verified"""

    verified = {}          # (pk, msg, sig) -> bool: the Python curve arithmetic takes 0.3 s per signature; the sets repeat a few

    def __init__(self, sigs_fn):
        self.sigs_fn, self.last, self.ok = sigs_fn, {}, {}

    def prove(self, ctx=None, job=None) -> bytes:
        raw, _ = self.sigs_fn(job)
        good, results = True, []
        for t in raw:
            if t not in HostBus.verified:
                HostBus.verified[t] = ea.verify(*t)
            good = good and HostBus.verified[t]
            results.append(ea.decompress(t[2][:32]))
        self.last[id(ctx)] = (raw, results, [[0]])
        self.ok[id(ctx)] = good
        return hashlib.sha256(b"bus|" + b"".join(pk + sig for pk, _, sig in raw)).digest()

    def closed(self, ctx=None) -> bool:
        return self.ok[id(ctx)]

    def describe(self):
        return "host verification (test stand-in)"

    ntab, cap = 1, 1 << 30

    def take_spent(self, ctx=None):
        return None

    def free(self):
        pass


class CpuTables:
    def hash_table(self, label, which, log_n, messages_fn, lanes):
        return HashlibTable(which, messages_fn), "hashlib"

    def signature_bus(self, sigs_fn, lanes, nsigs, sha_log_n, ed_log_n):
        return HostBus(sigs_fn)


class TablesProver:
    """GpuProver's contract without the GPU: a job = a stand-in main proof + its tables + its statement (mapreduce.prove_with_tables)"""
    takes_input_seed = True
    takes_children = True

    def __init__(self, kind, starks):
        self.kind, self.starks = kind, list(starks)
        self.emits_statement = any(getattr(t, "needs_children", False) for _, t in self.starks)
        self.lanes = [object() for _ in range(8)]

    def prove(self, key, public_inputs, lane=0, input_seed=b"", spent_out=None, with_tables=True, children=()):
        main = hashlib.sha256(b"|".join([self.kind.encode(), b"%d,%d" % tuple(key), bytes(public_inputs.tobytes()), bytes(input_seed)])).digest()
        return mr.prove_with_tables(lambda: main, self.starks if with_tables else (), self.lanes[lane], None, None,
                                    job=(self.kind, key[0], key[1], input_seed, tuple(children)), spent_out=spent_out)

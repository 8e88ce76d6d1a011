"""A stand-in prover for the scheduler tests of vectorx_amd/dag_pool.py (no GPU, no oracle): a job's "proof" is a hash of everything
that must determine it — kind, position, public inputs, request seed — and takes a few milliseconds."""
import hashlib
import time


class FakeProver:
    takes_input_seed = True

    def __init__(self, kind, delay=0.002):
        self.kind, self.delay = kind, delay

    def prove(self, key, public_inputs, lane=0, input_seed=b"", spent_out=None, with_tables=True):
        t0 = time.perf_counter()
        time.sleep(self.delay * (1 + (key[1] % 3)))
        proof = hashlib.sha256(b"|".join([self.kind.encode(), b"%d,%d" % tuple(key), bytes(public_inputs.tobytes()), bytes(input_seed)])).digest() * 4
        if with_tables:
            proof += self._tables(key, input_seed)
        if spent_out is not None:
            spent_out.append(("plonky2", time.perf_counter() - t0))
            if with_tables:
                spent_out.append(("trace_generation", 0.0001))
        return proof

    def _tables(self, key, input_seed):
        return hashlib.sha256(b"tables|" + self.kind.encode() + b"%d,%d" % tuple(key) + bytes(input_seed)).digest()

    def prove_tables(self, key, lane=0, input_seed=b"", spent_out=None):
        time.sleep(self.delay * 3)
        if spent_out is not None:
            spent_out.append(("eddsa", self.delay * 3))
        return self._tables(key, input_seed)


class StatingProver(FakeProver):
    """a prover that STATES something (mapreduce.with_statement / record_of) and reads what its children stated (`takes_children`): a
    map job states its index, a reduce job the hash of its children's statements, the outer job its child's statement reversed"""
    takes_children = True
    emits_statement = True

    def _statement(self, key, children):
        import struct
        if self.kind == "map":
            return struct.pack("<I", key[1]) * 2
        kids = [bytes(r)[32:] for r in children]
        assert all(len(k) > 0 for k in kids) and len(kids) == (1 if self.kind == "outer" else 2), (self.kind, key, [len(bytes(r)) for r in children])
        return kids[0][::-1] if self.kind == "outer" else hashlib.sha256(b"".join(kids)).digest()

    def prove(self, key, public_inputs, lane=0, input_seed=b"", spent_out=None, with_tables=True, children=()):
        proof = super().prove(key, public_inputs, lane, input_seed, spent_out, with_tables)
        return proof + self.prove_statement(key, lane, input_seed, children) if with_tables else proof

    def prove_statement(self, key, lane=0, input_seed=b"", children=(), spent_out=None):
        from vectorx_amd import mapreduce as mr
        return mr.with_statement(self._statement(key, children))


def make_stating(cfg, device):
    kinds = ("map", "reduce", "outer") if cfg["worker_index"] == 0 else ("map", "reduce")
    seen = []
    return {k: StatingProver(k) for k in kinds}, (lambda: None), {"fake": True, "_preload": seen.append}


def make(cfg, device):
    kinds = ("map", "reduce", "outer") if cfg["worker_index"] == 0 else ("map", "reduce")
    return {k: FakeProver(k) for k in kinds}, (lambda: None), {"fake": True}


def make_failing(cfg, device):
    provers, close, info = make(cfg, device)

    class Boom(FakeProver):
        def prove(self, key, *a, **kw):
            if key == (1, 1):
                raise RuntimeError("boom in reduce job 1")
            return super().prove(key, *a, **kw)
    provers["reduce"] = Boom("reduce")
    return provers, close, info


def make_dying(cfg, device):
    """worker 1 dies during its setup (the coordinator must notice within seconds, not wait for a timeout)"""
    if cfg["worker_index"] == 1:
        import os
        os._exit(7)
    return make(cfg, device)

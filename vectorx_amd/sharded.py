"""Sharding ONE proof across the G GPUs of one node (BASELINE.json configs[3]).

Two things live here:

* `prove_sharded_processes` / `prove_sharded_threads` — the whole proof, split by LDE COSET (`vx_prove_sharded`,
  include/vxprover.h): rank r owns LDE rows [r*N/G, (r+1)*N/G) of every committed polynomial, which in this library's
  bit-reversed row order are whole cosets, so coset NTTs, leaf hashing, Merkle subtrees, the quotient evaluation and
  the first FRI layer are rank-local and the ranks only all-gather small buffers (cap entries, 2*8n quotient coset
  coefficients, N/16 folded FRI values, query openings).  No LDE-sized exchange is needed at all, because the
  coefficients (n per column) are cheap to replicate and each rank extends them to ITS cosets only.
* `commit_sharded` — the row-chunk formulation BASELINE.json configs[3] names (column-sharded LDE, then ONE
  all-to-all that re-shards the LDE by row); kept as the L2 building block for hosts that hold the columns
  distributed, and as the comparison point: it moves the whole LDE (18 GB at n = 2^21) where the coset split moves
  0.3 GB.

---- commit_sharded: sharded `PolynomialBatch::from_values` ----

The commitment has exactly ONE exchange step:

  1. column-shard  — rank g owns a contiguous block of columns, interpolates them (iNTT) and extends them to the
                     coset LDE locally: columns are independent polynomials, no communication;
  2. all-to-all    — the LDE is re-sharded by ROW: rank h receives, from every rank, the rows
                     [h*N/G, (h+1)*N/G) of that rank's columns.  In the bit-reversed row order used everywhere in
                     this library those rows are the TOP log2(G) bits of the Merkle leaf index, so a
                     rank ends up with whole cap subtrees, and the quotient's "next row" neighbour stays local;
  3. row-shard     — leaf hashing (all columns of the local rows) + Merkle subtrees, no communication;
  4. all-gather    — of the 2^cap_height / G cap digests per rank (tiny).

`torch.distributed` supplies the collectives (backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU
tests); per-link volume at n = 2^21, 135 columns: 18.1 GB * (G-1)/G^2 per rank pair, i.e. 1.13 GB per link at
G = 4 and 0.28 GB at G = 8 — every one of the 7 xGMI links is used once, concurrently (point-to-point, not a ring).

The compute steps are delegated to a backend: `GpuBackend` (libvxprover.so kernels on torch-owned device buffers) or,
in the CPU tests only, an oracle-backed stand-in defined in tests/ — this module never imports the oracle.
"""
from __future__ import annotations

import numpy as np


def column_blocks(ncols: int, world: int):
    """Contiguous column blocks, padded so every rank owns the same count (padding columns are all-zero
    polynomials that are never hashed)."""
    per = (ncols + world - 1) // world
    return per, [(min(ncols, g * per), min(ncols, (g + 1) * per)) for g in range(world)]


class GpuBackend:
    """Kernels of libvxprover.so on torch CUDA tensors (int64 views of u64 data)."""

    def __init__(self, ctx, device):
        import torch
        self.ctx, self.torch, self.device = ctx, torch, device

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int64, device=self.device)

    def from_host(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(self.device)

    def lde_columns(self, values, log_n, rate_bits):
        """values [mc][n] (device) -> lde [mc][N] bit-reversed rows"""
        mc = values.shape[0]
        out = self.empty(mc, (1 << log_n) << rate_bits)
        self.torch.cuda.synchronize()
        self.ctx.lde_columns_dev(values.data_ptr(), log_n, mc, rate_bits, out.data_ptr())
        return out

    def hash_rows(self, rows, ncols, cap_height):
        """rows [m_padded][Nb] column-major (device); hashes the first ncols columns of every row"""
        self.torch.cuda.synchronize()
        return self.ctx.hash_rows_dev(rows.data_ptr(), rows.shape[1], rows.shape[1], ncols, cap_height)


def commit_sharded(backend, dist, local_values, ncols_total: int, log_n: int, rate_bits: int = 3, cap_height: int = 4):
    """local_values: this rank's column block [per][n] (natural row order; padded with zero columns to `per`).
    Returns (cap [2^cap_height][4] — identical on every rank, row_shard [world*per][N/world]).
    """
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lg = 0
    while (1 << lg) < world:
        lg += 1
    if (1 << lg) != world or lg > cap_height:
        raise ValueError("world size must be a power of two <= 2^cap_height")
    per, _ = column_blocks(ncols_total, world)
    if local_values.shape[0] != per:
        raise ValueError(f"rank {rank}: expected {per} local columns, got {local_values.shape[0]}")
    N = (1 << log_n) << rate_bits
    Nb = N // world
    # 1. column-shard: local iNTT + coset LDE
    lde = backend.lde_columns(local_values, log_n, rate_bits)                    # [per][N]
    # 2. one all-to-all: send block h = rows [h*Nb, (h+1)*Nb) of my columns; receive my rows of everyone's columns
    if world > 1:
        send = lde.reshape(per, world, Nb).permute(1, 0, 2).contiguous()           # [world][per][Nb]
        recv = backend.empty(world, per, Nb)
        dist.all_to_all_single(recv.view(-1), send.view(-1))
        rows = recv.reshape(world * per, Nb)                                       # column-major, columns in global order
    else:
        rows = lde
    # 3. row-shard: leaf hashing + subtrees down to this rank's share of the cap
    local_cap = backend.hash_rows(rows, ncols_total, cap_height - lg)              # [2^(cap_height-lg)][4] numpy u64
    # 4. all-gather of the cap digests (rank order == leaf-block order)
    if world > 1:
        import torch
        mine = torch.from_numpy(local_cap.view(np.int64).copy())
        gathered = [torch.empty_like(mine) for _ in range(world)]
        if rows.is_cuda:
            mine_d = mine.to(rows.device)
            gathered = [torch.empty_like(mine_d) for _ in range(world)]
            dist.all_gather(gathered, mine_d)
            gathered = [t.cpu() for t in gathered]
        else:
            dist.all_gather(gathered, mine)
        cap = np.concatenate([t.numpy().view(np.uint64) for t in gathered], axis=0)
    else:
        cap = local_cap
    return cap, rows


# ---------------------------------------------------------------------------------------------------------------
# Whole-proof sharding (vx_prove_sharded)
# ---------------------------------------------------------------------------------------------------------------
class _DeviceView:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchAllGather:
    """The in-place all-gather `vx_prove_sharded` asks its host for, on `torch.distributed`:
    backend "nccl" (= RCCL over xGMI) gathers straight into the library's device buffer; any other backend (gloo in
    the tests) stages the slots through host memory."""

    def __init__(self, ctx, dist, device=None):
        import torch
        self.ctx, self.dist, self.torch = ctx, dist, torch
        self.device = device
        self.calls, self.bytes = 0, 0
        self._ext = None

    def _library_stream(self):
        """the context's own HIP stream as a torch stream: a collective issued under it is ordered ON THE DEVICE after the kernels the
        library launched before the exchange and before the ones it launches after — no host synchronisation on either side"""
        if self._ext is None:
            self._ext = self.torch.cuda.ExternalStream(self.ctx.stream, device=self.device)
        return self._ext

    def __call__(self, dptr: int, nbytes: int):
        dist, torch = self.dist, self.torch
        world, rank = dist.get_world_size(), dist.get_rank()
        self.calls += 1
        self.bytes += nbytes * (world - 1)
        if dist.get_backend() == "nccl":
            # in place: the send buffer IS this rank's slot of the receive buffer (RCCL's in-place all-gather: sendbuff == recvbuff +
            # rank * count) — no staging copy, no device-wide synchronise (round 5; rounds 2-4 cloned the slot and synchronised the device
            # seven times per proof)
            full = torch.as_tensor(_DeviceView(dptr, nbytes * world), device=self.device)
            mine = full[rank * nbytes:(rank + 1) * nbytes]
            with torch.cuda.stream(self._library_stream()):
                dist.all_gather_into_tensor(full, mine)
            return
        mine = torch.from_numpy(self.ctx.download(dptr + rank * nbytes, nbytes).view(np.uint8).copy())
        slots = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(slots, mine)
        for r, t in enumerate(slots):
            if r != rank:
                self.ctx.upload(dptr + r * nbytes, t.numpy())


def prove_sharded_processes(circuit, wires, dist, device=None, pow_witness=None) -> bytes:
    """One process per GPU (`torch.distributed` already initialised): returns the proof on every rank."""
    world = dist.get_world_size() if dist is not None else 1
    if world == 1:
        return circuit.prove(wires, pow_witness=pow_witness)
    ag = TorchAllGather(circuit.ctx, dist, device)
    return circuit.prove_sharded(wires, dist.get_rank(), world, ag, pow_witness=pow_witness)


def prove_sharded_threads(circuits, wires, pow_witness=None, timeout_ms=600_000):
    """One process, one host thread + one `vx_ctx` (GPU) per rank, exchanging through the library's own `vx_group`
    (peer copies over xGMI) — the shape a Rust host with 8 worker threads uses.  `circuits[r]` is rank r's copy of
    the circuit (its own context).  Returns the list of per-rank proofs (all identical)."""
    import ctypes
    import threading

    from . import VxError, _chk, lib

    world = len(circuits)
    L = lib()
    g = ctypes.c_void_p()
    _chk(L.vx_group_create(world, ctypes.byref(g)))
    # ranks can reach their FIRST exchange far apart (one still uploading its share of the witness, or compiling the circuit's program
    # gates on first use): the harness sets the barrier timeout explicitly instead of inheriting the library's 120 s default
    _chk(L.vx_group_set_timeout_ms(g, int(timeout_ms)))
    out, errs = [None] * world, [None] * world
    try:
        members = []
        for r, c in enumerate(circuits):
            m = ctypes.c_void_p()
            _chk(L.vx_group_join(g, r, c.ctx._h, ctypes.byref(m)))
            members.append(m)

        def run(r):
            try:
                out[r] = circuits[r].prove_sharded(wires, r, world, L.vx_group_allgather, members[r], pow_witness=pow_witness)
            except BaseException as e:
                errs[r] = e
                L.vx_group_abort(g)

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        L.vx_group_destroy(g)
    real = [e for e in errs if e is not None and not (isinstance(e, VxError) and "aborted" in str(e))]
    if real or any(errs):
        raise (real or [e for e in errs if e is not None])[0]
    return out


def prove_stark_sharded_threads(ctxs, stark, trace, public_inputs, timeout_ms=600_000):
    """ONE STARK proof over len(ctxs) ranks (one host thread + one vx_ctx each, exchanging through `vx_group`): the coset split of
    vx_stark_begin_sharded.  Every rank computes the second-round columns itself (caller-side witness generation, replicated).
    Returns the per-rank proofs (all identical, and identical to `stark.prove`)."""
    import ctypes
    import threading

    from . import _chk, lib

    world = len(ctxs)
    L = lib()
    g = ctypes.c_void_p()
    _chk(L.vx_group_create(world, ctypes.byref(g)))
    _chk(L.vx_group_set_timeout_ms(g, int(timeout_ms)))
    out, errs = [None] * world, [None] * world
    try:
        members = []
        for r, c in enumerate(ctxs):
            m = ctypes.c_void_p()
            _chk(L.vx_group_join(g, r, c._h, ctypes.byref(m)))
            members.append(m)

        def run(r):
            sess = None
            try:
                sess, chal, t = stark.begin_sharded(ctxs[r], trace, public_inputs, r, world, L.vx_group_allgather, members[r])
                aux, api = (stark.run_aux(t, chal) if stark.desc.num_aux_columns else (np.zeros((0, 0), dtype=np.uint64), None))
                out[r] = stark.finish(sess, aux, api)
            except BaseException as e:
                errs[r] = e
                L.vx_group_abort(g)
            finally:
                if sess is not None:
                    L.vx_stark_session_free(sess)

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t_ in threads:
            t_.start()
        for t_ in threads:
            t_.join()
    finally:
        L.vx_group_destroy(g)
    if any(errs):
        raise [e for e in errs if e is not None][0]
    return out

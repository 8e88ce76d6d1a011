"""Sharded `PolynomialBatch::from_values` across the G GPUs of one node (BASELINE.json configs[3]).

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

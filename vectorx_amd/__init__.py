"""vectorx_amd — MI355X (gfx950) prover backend for VectorX's plonky2x circuits.

The product is ``libvxprover.so`` (hand-written HIP kernels behind the C ABI of ``include/vxprover.h``).
This package is the thin Python host-side mirror of the plonky2 surface that the reference reaches
through ``circuit.prove(&input)`` (/root/reference/circuits/header_range.rs:167) — same names and
argument meaning as plonky2 v0.2.0's ``PolynomialBatch::from_values``, ``MerkleTree::new``,
``fft`` / ``ifft`` / ``coset_fft`` — and it is what the parity tests and ``bench.py`` drive.

There is NO CPU fallback: importing works anywhere (so the symbol-export test can run on a CPU box),
but every compute call raises :class:`VxError` unless the HIP library is built and a gfx950 device is
present.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_LIB_PATH = Path(os.environ.get("VXPROVER_LIB", _PKG / "libvxprover.so"))  # override = kernel A/B experiments only
P = 0xFFFFFFFF00000001

VX_OK, VX_E_INVALID, VX_E_NO_DEVICE, VX_E_HIP, VX_E_NOMEM, VX_E_PROOF, VX_E_COMM = 0, -1, -2, -3, -4, -5, -6
VX_GATE_NOOP, VX_GATE_CONSTANT, VX_GATE_PUBLIC_INPUT, VX_GATE_ARITHMETIC, VX_GATE_POSEIDON, VX_GATE_PROGRAM = 0, 1, 2, 3, 4, 5
NTT_FFT, NTT_IFFT, NTT_COSET_FFT, NTT_COSET_IFFT = 0, 1, 2, 3


class VxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vxprover error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> Path:
    """Compile libvxprover.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", str(_PKG / "csrc")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libvxprover.so failed")
    return _LIB_PATH


_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_i = ctypes.c_int
_u64 = ctypes.c_uint64

# name -> (restype, argtypes).  Mirrors include/vxprover.h one-to-one (tests/test_abi.py checks that).
_SIGNATURES = {
    "vx_last_error": (ctypes.c_char_p, []),
    "vx_version": (ctypes.c_char_p, []),
    "vx_device_count": (_i, []),
    "vx_device_max_clock_khz": (_i, [_i]),
    "vx_ctx_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "vx_ctx_destroy": (None, [_vp]),
    "vx_ctx_sync": (_i, [_vp]),
    "vx_ctx_trim": (_i, [_vp]),
    "vx_ctx_stream": (_vp, [_vp]),
    "vx_clock_probe": (_i, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "vx_prof_enable": (_i, [_vp, _i]),
    "vx_prof_reset": (_i, [_vp]),
    "vx_prof_count": (_i, [_vp]),
    "vx_prof_get": (_i, [_vp, _i, ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_double), _u64p,
                         ctypes.POINTER(ctypes.c_double)]),
    "vx_dev_alloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "vx_dev_free": (_i, [_vp, _vp]),
    "vx_host_alloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "vx_host_free": (_i, [_vp, _vp]),
    "vx_dev_upload": (_i, [_vp, _vp, _vp, _sz]),
    "vx_dev_download": (_i, [_vp, _vp, _vp, _sz]),
    "vx_dev_upload_strided": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vx_ntt_batch": (_i, [_vp, _vp, _i, _sz, _i, _u64]),
    "vx_ntt_batch_dev": (_i, [_vp, _vp, _vp, _i, _sz, _i, _u64]),
    "vx_poseidon_permute": (_i, [_vp, _vp, _sz]),
    "vx_merkle_cap": (_i, [_vp, _vp, _sz, _sz, _i, _vp, _vp]),
    "vx_field_op": (_i, [_vp, _i, _vp, _vp, _vp, _sz]),
    "vx_batch_commit": (_i, [_vp, _vp, _i, _i, _sz, _i, _i, _i, ctypes.POINTER(_vp)]),
    "vx_batch_free": (None, [_vp]),
    "vx_batch_cap": (_i, [_vp, _vp]),
    "vx_batch_coeffs": (_i, [_vp, _sz, _vp]),
    "vx_batch_open_row": (_i, [_vp, _sz, _vp, _vp]),
    "vx_batch_digests": (_i, [_vp, _vp]),
    "vx_batch_lde_rows": (_i, [_vp, _sz, _sz, _vp]),
    "vx_batch_eval_ext": (_i, [_vp, _vp, _vp]),
    "vx_lde_columns_dev": (_i, [_vp, _vp, _i, _sz, _i, _vp, _vp]),
    "vx_hash_rows_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _i, _vp, _vp]),
    "vx_merkle_digest_count": (_sz, [_sz, _i]),
    "vx_circuit_create": (_i, [_vp, _vp, ctypes.POINTER(_vp)]),
    "vx_circuit_free": (None, [_vp]),
    "vx_circuit_warm": (_i, [_vp, _vp]),
    "vx_circuit_serialized_size": (_sz, [_vp, _i, _i]),
    "vx_circuit_serialize": (_i, [_vp, _vp, _i, _vp, ctypes.POINTER(_sz)]),
    "vx_circuit_parse": (_i, [_vp, _sz, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "vx_circuit_desc_free": (None, [_vp]),
    "vx_circuit_load": (_i, [_vp, _vp, _sz, ctypes.POINTER(_vp)]),
    "vx_circuit_digest": (_i, [_vp, _vp]),
    "vx_circuit_constants_sigmas_cap": (_i, [_vp, _vp]),
    "vx_prove": (_i, [_vp, _vp, _vp, _i, _vp, _vp, ctypes.POINTER(_sz)]),
    "vx_proof_size_bound": (_sz, [_vp]),
    "vx_verify": (_i, [_vp, _vp, _sz]),
    "vx_verify_standalone": (_i, [_vp, _vp, _vp, _sz]),
    "vx_stark_prove": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, ctypes.POINTER(_sz)]),
    "vx_stark_verify": (_i, [_vp, _vp, _vp, _sz]),
    "vx_stark_begin": (_i, [_vp, _vp, _vp, _i, _vp, _vp, ctypes.POINTER(_vp)]),
    "vx_stark_finish": (_i, [_vp, _vp, _i, _vp, _vp, ctypes.POINTER(_sz)]),
    "vx_stark_begin_sharded": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, ctypes.POINTER(_vp)]),
    "vx_stark_session_free": (None, [_vp]),
    "vx_stark_aux_columns": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "vx_trace_sha256": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vx_trace_sha512": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vx_trace_blake2b": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vx_trace_sha512_bus": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vx_trace_eddsa": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "vx_stark_aux_precompile": (_i, [_vp]),
    "vx_stark_precompile": (_i, [_vp, ctypes.POINTER(_i)]),
    "vx_circuit_precompile": (_i, [_vp, ctypes.POINTER(_i)]),
    "vx_stark_session_trace_cap": (_i, [_vp, _vp]),
    "vx_stark_set_aux_challenges": (_i, [_vp, _vp]),
    "vx_stark_finish2": (_i, [_vp, _vp, _i, _vp, _vp, _vp, ctypes.POINTER(_sz)]),
    "vx_stark_joint_challenges": (_i, [_vp, _vp, _i, _i, _vp]),
    "vx_stark_proof_trace_cap": (_i, [_vp, _vp, _sz, _vp]),
    "vx_stark_verify_shared": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "vx_stark_verify_bus": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "vx_circuit_program_gates": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.c_char_p, _sz]),
    "vx_prove_sharded": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, ctypes.POINTER(_sz)]),
    "vx_group_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "vx_group_destroy": (None, [_vp]),
    "vx_group_join": (_i, [_vp, _i, _vp, ctypes.POINTER(_vp)]),
    "vx_group_peer_staged": (_i, [_vp]),
    "vx_group_allgather": (_i, [_vp, _vp, _sz]),
    "vx_group_abort": (None, [_vp]),
    "vx_group_set_timeout_ms": (_i, [_vp, ctypes.c_longlong]),
}
# vx_allgather_fn: int (*)(void* user, void* dev_buf, size_t bytes_per_rank)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


def lib() -> ctypes.CDLL:
    """Load libvxprover.so.  Fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise VxError(VX_E_NO_DEVICE, f"{_LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc, gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if it is loaded AFTER the
        # system one that libvxprover.so would pull in, torch.cuda later reports "No HIP GPUs are available".
        # Importing torch first makes both resolve to the same runtime (torch is only plumbing here: device
        # buffers for the multi-GPU path, torch.distributed).  Without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = ctypes.CDLL(str(_LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _chk(rc: int):
    if rc != 0:
        raise VxError(rc, lib().vx_last_error().decode(errors="replace"))


def _as_u64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One (device, HIP stream) pair — `vx_ctx`."""

    def __init__(self, device: int = 0):
        import weakref
        self._h = _vp()
        _chk(lib().vx_ctx_create(device, ctypes.byref(self._h)))
        self.device = device
        # handles created on this context (Circuit, PolynomialBatch): vx_ctx_destroy releases the pool blocks they point
        # into, so close() frees the live ones FIRST — a later child.free() / __del__ then finds a null handle (no-op)
        self._children = weakref.WeakSet()

    def _adopt(self, child):
        self._children.add(child)

    def close(self):
        if self._h:
            for child in list(self._children):
                try:
                    child.free()
                except Exception:
                    pass
            lib().vx_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _chk(lib().vx_ctx_sync(self._h))

    def trim(self):
        """return the cached device buffers of this context's pool to the driver (live objects stay)"""
        _chk(lib().vx_ctx_trim(self._h))

    @property
    def stream(self) -> int:
        return lib().vx_ctx_stream(self._h)

    def clock_ghz(self) -> float:
        """effective shader clock under a VALU-saturating load, measured on the device (vx_clock_probe)"""
        g = ctypes.c_double()
        _chk(lib().vx_clock_probe(self._h, ctypes.byref(g)))
        return g.value

    # ---- profiling ----
    def prof_enable(self, on: bool = True):
        _chk(lib().vx_prof_enable(self._h, int(on)))

    def prof_reset(self):
        _chk(lib().vx_prof_reset(self._h))

    def prof(self) -> dict:
        n = lib().vx_prof_count(self._h)
        out = {}
        for k in range(n):
            name = ctypes.create_string_buffer(64)
            ms, by, calls = ctypes.c_double(), ctypes.c_double(), _u64()
            _chk(lib().vx_prof_get(self._h, k, name, 64, ctypes.byref(ms), ctypes.byref(calls), ctypes.byref(by)))
            out[name.value.decode()] = {"ms": ms.value, "calls": calls.value, "alg_bytes": by.value}
        return out

    # ---- chip-table traces generated on the device (vx_trace_*) ----
    TRACE_TABLES = {"sha256": (1024, 8, 32), "sha512": (1995, 16, 64), "blake2b": (775, 8, 32),
                    "sha512_bus": (2012, 16, 64)}                                                   # columns, public inputs, digest bytes

    def trace_hash_table(self, which: str, degree_bits: int, messages, d_trace: int):
        """Fill the `which` table ("sha256" | "sha512" | "blake2b": the AIRs of vectorx_amd/{sha256,sha512,blake2b_bytes}_air.py) for
        `messages` (byte strings, hashed one after the other) into device memory d_trace = [columns][2^degree_bits]
        -> (public inputs of the table's AIR, [digest bytes per message])."""
        ncols, npis, dlen = self.TRACE_TABLES[which]
        blob = b"".join(messages)
        off = np.zeros(len(messages) + 1, dtype=np.uint64)
        np.cumsum([len(m) for m in messages], out=off[1:])
        buf = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, dtype=np.uint8)
        pis = np.zeros(npis, dtype=np.uint64)
        dig = np.zeros(max(1, dlen * len(messages)), dtype=np.uint8)
        _chk(getattr(lib(), "vx_trace_" + which)(self._h, degree_bits, buf.ctypes.data, off.ctypes.data, len(messages), _vp(d_trace),
                                                 pis.ctypes.data, dig.ctypes.data))
        return pis, [dig[dlen * i:dlen * (i + 1)].tobytes() for i in range(len(messages))]

    def trace_eddsa_table(self, degree_bits: int, scalar_bits: int, sigs, d_trace: int, full: bool = False):
        """Fill the batched EdDSA table (eddsa_air.Layout(16, scalar_bits, full)) for sigs = [((ax, ay), S, h)] — full: [((ax, ay), S, h,
        digest)] — into device memory d_trace -> [(x, y)] per signature = the affine [S]B - [h]A the instance arrives at (Python integers)."""
        stride = 6 if full else 4
        arr = np.zeros((max(1, len(sigs)), stride, 4), dtype=np.uint64)
        words = lambda v: [(int(v) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)]      # noqa: E731
        for i, sg in enumerate(sigs):
            (ax, ay), s_, h_ = sg[0], sg[1], sg[2]
            arr[i, 0], arr[i, 1], arr[i, 2], arr[i, 3] = words(ax), words(ay), words(s_), words(h_)
            if full:
                arr[i, 4], arr[i, 5] = words(sg[3]), words(int(sg[3]) >> 256)
        res = np.zeros((max(1, len(sigs)), 2, 4), dtype=np.uint64)
        _chk(lib().vx_trace_eddsa(self._h, degree_bits, scalar_bits, 1 if full else 0, arr.ctypes.data, len(sigs), _vp(d_trace), res.ctypes.data))
        val = lambda w: sum(int(w[k]) << (64 * k) for k in range(4))      # noqa: E731
        return [(val(res[i, 0]), val(res[i, 1])) for i in range(len(sigs))]

    # ---- device buffers ----
    def alloc(self, nbytes: int) -> int:
        p = _vp()
        _chk(lib().vx_dev_alloc(self._h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, dptr: int):
        _chk(lib().vx_dev_free(self._h, dptr))

    def host_alloc(self, shape) -> np.ndarray:
        """page-locked uint64 host array (freed with host_free(arr))"""
        n = int(np.prod(shape))
        p = _vp()
        _chk(lib().vx_host_alloc(self._h, n * 8, ctypes.byref(p)))
        buf = (ctypes.c_uint64 * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint64).reshape(shape)
        return arr

    def host_free(self, arr: np.ndarray):
        _chk(lib().vx_host_free(self._h, arr.ctypes.data))

    def upload(self, dptr: int, host: np.ndarray):
        host = np.ascontiguousarray(host)
        _chk(lib().vx_dev_upload(self._h, dptr, host.ctypes.data, host.nbytes))

    def upload_row(self, dptr: int, n_rows: int, row: int, values: np.ndarray):
        """write one row of a column-major [len(values)][n_rows] u64 matrix that lives at dptr"""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        _chk(lib().vx_dev_upload_strided(self._h, dptr + 8 * row, 8 * n_rows, v.ctypes.data, v.size))

    def download(self, dptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes // 8, dtype=np.uint64)
        _chk(lib().vx_dev_download(self._h, out.ctypes.data, dptr, nbytes))
        return out

    def download_into(self, host: np.ndarray, dptr: int):
        """device -> an existing (e.g. page-locked) contiguous host array"""
        assert host.flags["C_CONTIGUOUS"]
        _chk(lib().vx_dev_download(self._h, host.ctypes.data, dptr, host.nbytes))

    # ---- L1 (plonky2_field::fft / hash::poseidon / hash::merkle_tree) ----
    def ntt_batch(self, cols: np.ndarray, kind: int, shift: int = 7) -> np.ndarray:
        """cols: [ncols][n] column-major; returns the transformed copy, natural order (fft.rs)."""
        a = _as_u64(cols).copy()
        if a.ndim == 1:
            a = a[None, :]
        ncols, n = a.shape
        log_n = int(n).bit_length() - 1
        if n == 0 or (1 << log_n) != n:
            raise VxError(VX_E_INVALID, f"length {n} is not a power of two")
        _chk(lib().vx_ntt_batch(self._h, a.ctypes.data, log_n, ncols, kind, shift))
        return a

    def ntt_batch_dev(self, src: int, dst: int, log_n: int, ncols: int, kind: int, shift: int = 7):
        _chk(lib().vx_ntt_batch_dev(self._h, src, dst, log_n, ncols, kind, shift))

    def lde_columns_dev(self, values_ptr: int, log_n: int, ncols: int, rate_bits: int, lde_ptr: int, coeffs_ptr: int = 0):
        _chk(lib().vx_lde_columns_dev(self._h, values_ptr, log_n, ncols, rate_bits, lde_ptr, coeffs_ptr or None))

    def hash_rows_dev(self, cols_ptr: int, col_stride: int, nrows: int, ncols: int, cap_height: int, tree_ptr: int = 0):
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        _chk(lib().vx_hash_rows_dev(self._h, cols_ptr, col_stride, nrows, ncols, cap_height, tree_ptr or None, cap.ctypes.data))
        return cap

    def field_op(self, op: int, a, b) -> np.ndarray:
        a, b = _as_u64(a), _as_u64(b)
        out = np.empty_like(a)
        _chk(lib().vx_field_op(self._h, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
        return out

    def poseidon_permute(self, states: np.ndarray) -> np.ndarray:
        a = _as_u64(states).reshape(-1, 12).copy()
        _chk(lib().vx_poseidon_permute(self._h, a.ctypes.data, a.shape[0]))
        return a

    def merkle_cap(self, leaves: np.ndarray, cap_height: int):
        """MerkleTree::new(leaves, cap_height) -> (digests [n][4], cap [2^cap_height][4])."""
        a = _as_u64(leaves)
        n, w = a.shape
        dig = np.empty((n, 4), dtype=np.uint64)
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        _chk(lib().vx_merkle_cap(self._h, a.ctypes.data, n, w, cap_height, dig.ctypes.data, cap.ctypes.data))
        return dig, cap


class PolynomialBatch:
    """Device-resident plonky2 `PolynomialBatch` (fri/oracle.rs) — `vx_batch`."""

    def __init__(self, ctx: Context, handle, log_n, ncols, rate_bits, cap_height):
        self.ctx, self._h = ctx, handle
        self.log_n, self.ncols, self.rate_bits, self.cap_height = log_n, ncols, rate_bits, cap_height
        ctx._adopt(self)

    @classmethod
    def _commit(cls, ctx, cols, rate_bits, cap_height, is_coeffs, dev_ptr=None, log_n=None, ncols=None):
        h = _vp()
        if dev_ptr is None:
            a = _as_u64(cols)
            ncols, n = a.shape
            log_n = int(n).bit_length() - 1
            if n == 0 or (1 << log_n) != n:
                raise VxError(VX_E_INVALID, f"length {n} is not a power of two")
            _chk(lib().vx_batch_commit(ctx._h, a.ctypes.data, 0, log_n, ncols, rate_bits, cap_height, is_coeffs,
                                       ctypes.byref(h)))
        else:
            _chk(lib().vx_batch_commit(ctx._h, dev_ptr, 1, log_n, ncols, rate_bits, cap_height, is_coeffs,
                                       ctypes.byref(h)))
        return cls(ctx, h, log_n, ncols, rate_bits, cap_height)

    @classmethod
    def from_values(cls, ctx, values, rate_bits=3, cap_height=4):
        return cls._commit(ctx, values, rate_bits, cap_height, 0)

    @classmethod
    def from_coeffs(cls, ctx, coeffs, rate_bits=3, cap_height=4):
        return cls._commit(ctx, coeffs, rate_bits, cap_height, 1)

    @classmethod
    def from_values_dev(cls, ctx, dev_ptr, log_n, ncols, rate_bits=3, cap_height=4):
        return cls._commit(ctx, None, rate_bits, cap_height, 0, dev_ptr=dev_ptr, log_n=log_n, ncols=ncols)

    def free(self):
        if self._h:
            lib().vx_batch_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def n(self):
        return 1 << self.log_n

    @property
    def lde_size(self):
        return 1 << (self.log_n + self.rate_bits)

    def cap(self) -> np.ndarray:
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        _chk(lib().vx_batch_cap(self._h, out.ctypes.data))
        return out

    def coeffs(self, col: int) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint64)
        _chk(lib().vx_batch_coeffs(self._h, col, out.ctypes.data))
        return out

    def digests(self) -> np.ndarray:
        out = np.empty((self.lde_size, 4), dtype=np.uint64)
        _chk(lib().vx_batch_digests(self._h, out.ctypes.data))
        return out

    def lde_rows(self, row0: int, nrows: int) -> np.ndarray:
        out = np.empty((nrows, self.ncols), dtype=np.uint64)
        _chk(lib().vx_batch_lde_rows(self._h, row0, nrows, out.ctypes.data))
        return out

    def open_row(self, row: int):
        """(leaf values, Merkle path) — `MerkleTree::prove(row)` + `tree.get(row)`."""
        vals = np.empty(self.ncols, dtype=np.uint64)
        path = np.empty((self.log_n + self.rate_bits - self.cap_height, 4), dtype=np.uint64)
        _chk(lib().vx_batch_open_row(self._h, row, vals.ctypes.data, path.ctypes.data))
        return vals, path

    def eval_ext(self, zeta) -> np.ndarray:
        z = _as_u64(zeta, (2,))
        out = np.empty((self.ncols, 2), dtype=np.uint64)
        _chk(lib().vx_batch_eval_ext(self._h, z.ctypes.data, out.ctypes.data))
        return out


def verify_standalone(desc_ptr, constants_sigmas_cap, proof: bytes) -> None:
    """`vx_verify_standalone`: CircuitData::verify from verifier data alone (no GPU, no context).  `desc_ptr` as for
    Circuit(), `constants_sigmas_cap` = [2^cap_height][4] u64.  Raises VxError(VX_E_PROOF, reason) on an invalid proof."""
    cap = _as_u64(constants_sigmas_cap)
    buf = np.frombuffer(proof, dtype=np.uint8) if len(proof) else np.zeros(1, np.uint8)
    _chk(lib().vx_verify_standalone(ctypes.cast(desc_ptr, _vp), cap.ctypes.data, buf.ctypes.data, len(proof)))


VX_STARK_OPENINGS_DIGEST = 8


class StarkDesc(ctypes.Structure):
    """ctypes mirror of `vx_stark_desc` (include/vxprover.h)."""
    _fields_ = [("degree_bits", ctypes.c_int32), ("num_columns", ctypes.c_int32), ("num_public_inputs", ctypes.c_int32),
                ("rate_bits", ctypes.c_int32), ("cap_height", ctypes.c_int32), ("pow_bits", ctypes.c_int32),
                ("num_query_rounds", ctypes.c_int32), ("num_challenges", ctypes.c_int32), ("constraint_degree", ctypes.c_int32),
                ("program_len", ctypes.c_int32), ("program", ctypes.c_void_p), ("override_flags", ctypes.c_uint32),
                ("num_fri_reduction_arity_bits", ctypes.c_int32), ("fri_reduction_arity_bits", ctypes.c_void_p),
                ("num_aux_columns", ctypes.c_int32), ("num_aux_challenges", ctypes.c_int32), ("num_aux_public_inputs", ctypes.c_int32)]


VX_OP_END, VX_OP_LDW, VX_OP_LDC, VX_OP_LDI, VX_OP_ADD, VX_OP_SUB, VX_OP_MUL, VX_OP_PUSH, VX_OP_LDP, VX_OP_LDN, VX_OP_LDCH = range(11)
VX_AIR_ALL_ROWS, VX_AIR_TRANSITION, VX_AIR_FIRST_ROW, VX_AIR_LAST_ROW = range(4)


def vx_ins(op, dst=0, a=0, b=0) -> int:
    """VX_INS of include/vxprover.h"""
    return op | (dst << 8) | (a << 16) | (b << 32)


def replicate_aux_program(program, num_columns, num_public_inputs, num_aux_columns, num_aux_challenges, num_aux_public_inputs, reps):
    """The second-round argument of an AIR program, repeated once per challenge set (what starky does with its lookups for each of
    `num_challenges`): a log-derivative lookup or a bus over ONE base-field challenge is sound to about (lookups + table size) / 2^64
    — ~2^-43 at 2^13 .. 2^18 rows — far below what the FRI configuration targets; with `reps` independent challenge sets, each with
    its own helper / accumulator columns and closing sums, the error is that to the power `reps`.

    `program` is written for ONE set.  Set 0 is the program as it stands; for set r >= 1 the backward slice of every constraint that
    depends on a challenge, a second-round column or a closing sum is emitted again (same instructions, same registers, same order —
    an instruction outside the slice never feeds one inside it) with challenge index + r * num_aux_challenges, second-round
    column + r * num_aux_columns, closing sum + r * num_aux_public_inputs.  Constraints of the first round are not repeated."""
    ins = []          # (op, dst, a, b, imm)
    i, words = 0, [int(w) for w in program]
    while i < len(words):
        w = words[i]
        op, dst, a, b = w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFFFF, (w >> 32) & 0xFFFFFFFF
        if op == VX_OP_END:
            break
        imm = None
        if op == VX_OP_LDI:
            i += 1
            imm = words[i]
        ins.append((op, dst, a, b, imm))
        i += 1
    if reps <= 1:
        return list(words)
    last_def = {}                 # register -> index of the instruction that wrote it last
    srcs, taint = [], []
    for k, (op, dst, a, b, imm) in enumerate(ins):
        if op in (VX_OP_ADD, VX_OP_SUB, VX_OP_MUL):
            s = [last_def[a], last_def[b]]
        elif op == VX_OP_PUSH:
            s = [last_def[a]]
        else:
            s = []
        own = (op == VX_OP_LDCH) or (op in (VX_OP_LDW, VX_OP_LDN) and a >= num_columns) or (op == VX_OP_LDP and a >= num_public_inputs)
        srcs.append(s)
        taint.append(own or any(taint[j] for j in s))
        if op != VX_OP_PUSH:
            last_def[dst] = k
    need = [False] * len(ins)
    stack = [k for k, t in enumerate(ins) if t[0] == VX_OP_PUSH and taint[k]]
    while stack:
        k = stack.pop()
        if need[k]:
            continue
        need[k] = True
        stack.extend(srcs[k])
    out = []

    def emit(op, dst, a, b, imm):
        out.append(vx_ins(op, dst, a, b))
        if imm is not None:
            out.append(imm)

    for t in ins:
        emit(*t)
    for r in range(1, reps):
        for k, (op, dst, a, b, imm) in enumerate(ins):
            if not need[k]:
                continue
            if op == VX_OP_LDCH:
                a += r * num_aux_challenges
            elif op in (VX_OP_LDW, VX_OP_LDN) and a >= num_columns:
                a += r * num_aux_columns
            elif op == VX_OP_LDP and a >= num_public_inputs:
                a += r * num_aux_public_inputs
            emit(op, dst, a, b, imm)
    out.append(vx_ins(VX_OP_END))
    return out


class Stark:
    """An AIR as a constraint program + its STARK configuration (`vx_stark_desc`).  Defaults = starky's
    StarkConfig::standard_fast_config (rate_bits 1, cap_height 4, 16 PoW bits, 84 queries, 2 challenges)."""

    def __init__(self, degree_bits, num_columns, num_public_inputs, program, constraint_degree, rate_bits=1, cap_height=4, pow_bits=16,
                 num_query_rounds=84, num_challenges=2, fri_arities=None, num_aux_columns=0, num_aux_challenges=0, aux_fn=None,
                 num_aux_public_inputs=0, aux_reps=None, openings_digest=False):
        """`openings_digest`: the transcript absorbs a tree hash of the opening set, computed on the device, instead of the opening set
        (`VX_STARK_OPENINGS_DIGEST`, include/vxprover.h) — for wide tables; prover and verifier must both set it.
        `num_aux_columns` / `num_aux_challenges` / `aux_fn`: a second commitment round — `aux_fn(trace, challenges)` returns the
        [num_aux_columns][n] columns the caller computes between `vx_stark_begin` and `vx_stark_finish`; with
        `num_aux_public_inputs` > 0 (closing sums of a bus / lookup accumulator) it returns `(columns, aux_public_inputs)`.
        These counts, the program and `aux_fn` describe ONE challenge set; the second round is repeated `aux_reps` times
        (default: `num_challenges`, as starky repeats its lookups) by `replicate_aux_program`, so the description the library sees has
        `aux_reps` times the columns, challenges and closing sums (set-major: everything of set 0, then set 1, ...)."""
        self.aux_reps = (num_challenges if aux_reps is None else aux_reps) if num_aux_columns > 0 else 1
        self.per_set = (num_aux_columns, num_aux_challenges, num_aux_public_inputs)
        if self.aux_reps > 1:
            program = replicate_aux_program(program, num_columns, num_public_inputs, num_aux_columns, num_aux_challenges,
                                            num_aux_public_inputs, self.aux_reps)
        self._prog = (ctypes.c_uint64 * len(program))(*program)
        self.desc = StarkDesc(degree_bits, num_columns, num_public_inputs, rate_bits, cap_height, pow_bits, num_query_rounds, num_challenges,
                              constraint_degree, len(program), ctypes.cast(self._prog, ctypes.c_void_p).value, 0, 0, None,
                              num_aux_columns * self.aux_reps, num_aux_challenges * self.aux_reps, num_aux_public_inputs * self.aux_reps)
        self.aux_fn = aux_fn
        self.aux_program = None          # an AuxProgram (one challenge set): the GPU form of aux_fn, set by the table's make_stark
        if fri_arities is not None:
            self._ar = (ctypes.c_int32 * max(1, len(fri_arities)))(*fri_arities)
            self.desc.override_flags = 2
            self.desc.num_fri_reduction_arity_bits = len(fri_arities)
            self.desc.fri_reduction_arity_bits = ctypes.cast(self._ar, ctypes.c_void_p).value
        if openings_digest:
            self.desc.override_flags |= VX_STARK_OPENINGS_DIGEST
        self.desc_ptr = ctypes.pointer(self.desc)

    def precompile(self) -> tuple:
        """`vx_stark_precompile`: compile every chunk of the AIR program with hiprtc (no GPU needed) -> (compiled now, chunks)"""
        n = _i(0)
        rc = lib().vx_stark_precompile(ctypes.cast(self.desc_ptr, _vp), ctypes.byref(n))
        if rc < 0:
            _chk(rc)
        return rc, n.value

    def run_aux_gpu(self, ctx, d_trace: int, challenges, d_aux: int) -> np.ndarray:
        """the second-round columns of every challenge set computed ON THE GPU (`self.aux_program`: an AuxProgram for one set) into the
        device buffer `d_aux` ([num_aux_columns][n]); -> the aux public inputs (closing sums), set-major like `run_aux`"""
        ap = self.aux_program
        naux, nch, _ = self.per_set
        n = 1 << self.desc.degree_bits
        if ap.num_out != naux:
            raise VxError(VX_E_INVALID, f"the aux program writes {ap.num_out} columns, the table has {naux} per challenge set")
        api = []
        for r in range(self.aux_reps):
            closing = ap.run(ctx, d_trace, self.desc.degree_bits, challenges[r * nch:(r + 1) * nch], d_aux + r * naux * n * 8)
            api.extend(int(closing[j]) for j in ap.api_sums)
        return np.array(api, dtype=np.uint64)

    def run_aux(self, trace, challenges):
        """-> (aux columns [num_aux_columns][n] uint64, aux public inputs [num_aux_public_inputs] uint64)"""
        nch = self.per_set[1]
        cols, api = [], []
        for r in range(self.aux_reps):           # one run of the table's aux_fn per challenge set
            res = self.aux_fn(trace, challenges[r * nch:(r + 1) * nch])
            c, a = res if isinstance(res, tuple) else (res, np.zeros(0, dtype=np.uint64))
            cols.append(_as_u64(c))
            api.append(np.ascontiguousarray(a, dtype=np.uint64).reshape(-1))
        cols, api = np.ascontiguousarray(np.concatenate(cols, axis=0)), np.concatenate(api)
        if cols.shape != (self.desc.num_aux_columns, 1 << self.desc.degree_bits) or api.size != self.desc.num_aux_public_inputs:
            raise VxError(VX_E_INVALID, f"aux columns have shape {cols.shape}, {api.size} aux public inputs")
        return cols, api

    # ---- two-round / multi-table building blocks (the pieces `prove` and vectorx_amd.stark_bus are made of) ----
    def begin(self, ctx, trace, public_inputs):
        """`vx_stark_begin` -> (session handle, the table's own aux challenges, trace as contiguous uint64)"""
        t = _as_u64(trace)
        if t.shape != (self.desc.num_columns, 1 << self.desc.degree_bits):
            raise VxError(VX_E_INVALID, f"trace has shape {t.shape}")
        pi = _as_u64(public_inputs)
        chal = np.zeros(max(1, self.desc.num_aux_challenges), dtype=np.uint64)
        sess = _vp()
        _chk(lib().vx_stark_begin(ctx._h, ctypes.cast(self.desc_ptr, _vp), t.ctypes.data, 0, pi.ctypes.data, chal.ctypes.data, ctypes.byref(sess)))
        return sess, chal[:self.desc.num_aux_challenges].copy(), t

    def begin_sharded(self, ctx, trace, public_inputs, rank: int, world: int, allgather, user=None):
        """`vx_stark_begin_sharded`: this rank's part of ONE proof split over `world` GPUs by LDE coset (allgather / user as in
        Circuit.prove_sharded: `lib().vx_group_allgather` + the member handle, or a Python callable).  Returns like `begin`; finish the
        session with `finish` — every rank gets the full proof, byte-identical to the unsharded one."""
        t = _as_u64(trace)
        if t.shape != (self.desc.num_columns, 1 << self.desc.degree_bits):
            raise VxError(VX_E_INVALID, f"trace has shape {t.shape}")
        pi = _as_u64(public_inputs)
        chal = np.zeros(max(1, self.desc.num_aux_challenges), dtype=np.uint64)
        sess = _vp()
        keep = None
        if callable(allgather) and not isinstance(allgather, ctypes._CFuncPtr):
            def _cb(_user, dptr, nbytes):
                try:
                    allgather(dptr, nbytes)
                    return 0
                except BaseException:   # never unwind through the C frames
                    return 1
            keep = ALLGATHER_FN(_cb)
            fn_ptr = ctypes.cast(keep, ctypes.c_void_p)
        else:
            fn_ptr = ctypes.cast(allgather, ctypes.c_void_p) if allgather is not None else None
        _chk(lib().vx_stark_begin_sharded(ctx._h, ctypes.cast(self.desc_ptr, _vp), t.ctypes.data, 0, pi.ctypes.data if pi.size else None, rank, world,
                                          fn_ptr, user, chal.ctypes.data, ctypes.byref(sess)))
        # the ctypes trampoline must outlive the session — one entry PER SESSION (several sharded sessions of one Stark may be open at once:
        # ranks as threads, overlapping proofs); dropped when the session has produced its proof (finish) or is freed (session_free)
        if not isinstance(getattr(self, "_keep_cb", None), dict):
            self._keep_cb = {}
        self._keep_cb[sess.value] = keep
        return sess, chal[:self.desc.num_aux_challenges].copy(), t

    def session_trace_cap(self, sess) -> np.ndarray:
        cap = np.zeros((1 << self.desc.cap_height, 4), dtype=np.uint64)
        _chk(lib().vx_stark_session_trace_cap(sess, cap.ctypes.data))
        return cap

    def finish(self, sess, aux, aux_public_inputs=None, pow_witness=None) -> bytes:
        cap = 1 << 25
        out = np.empty(cap, dtype=np.uint8)
        n = _sz(cap)
        hint = ctypes.c_uint64(pow_witness) if pow_witness is not None else None
        hint_p = None if hint is None else ctypes.cast(ctypes.pointer(hint), _vp)
        api = None if aux_public_inputs is None or len(aux_public_inputs) == 0 else np.ascontiguousarray(aux_public_inputs, dtype=np.uint64)
        try:
            _chk(lib().vx_stark_finish2(sess, aux.ctypes.data, 0, None if api is None else api.ctypes.data, hint_p, out.ctypes.data, ctypes.byref(n)))
        finally:
            if isinstance(getattr(self, "_keep_cb", None), dict):     # the session makes no exchange after its proof is out
                self._keep_cb.pop(getattr(sess, "value", None), None)
        return out[:n.value].tobytes()

    def prove(self, ctx, trace, public_inputs, pow_witness=None) -> bytes:
        """`vx_stark_prove`: trace [num_columns][2^degree_bits] (host) -> proof bytes"""
        if self.desc.num_aux_columns == 0:
            t = _as_u64(trace)
            if t.shape != (self.desc.num_columns, 1 << self.desc.degree_bits):
                raise VxError(VX_E_INVALID, f"trace has shape {t.shape}")
            pi = _as_u64(public_inputs)
            cap = 1 << 24
            out = np.empty(cap, dtype=np.uint8)
            n = _sz(cap)
            hint = ctypes.c_uint64(pow_witness) if pow_witness is not None else None
            hint_p = None if hint is None else ctypes.cast(ctypes.pointer(hint), _vp)
            _chk(lib().vx_stark_prove(ctx._h, ctypes.cast(self.desc_ptr, _vp), t.ctypes.data, 0, pi.ctypes.data, hint_p, out.ctypes.data, ctypes.byref(n)))
            return out[:n.value].tobytes()
        # two rounds: commit the trace, get the challenges, compute the aux columns (and closing sums), finish
        sess, chal, t = self.begin(ctx, trace, public_inputs)
        try:
            aux, api = self.run_aux(t, chal)
            return self.finish(sess, aux, api, pow_witness)
        finally:
            lib().vx_stark_session_free(sess)

    def verify(self, public_inputs, proof: bytes, shared_challenges=None):
        """`vx_stark_verify` (host): raises VxError(VX_E_PROOF, reason) when the proof is not valid; returns the proof's aux
        public inputs (closing sums).  `shared_challenges`: the joint challenges of a cross-table argument."""
        pi = _as_u64(public_inputs)
        buf = np.frombuffer(proof, dtype=np.uint8)
        api = np.zeros(max(1, self.desc.num_aux_public_inputs), dtype=np.uint64)
        sh = None if shared_challenges is None else np.ascontiguousarray(shared_challenges, dtype=np.uint64)
        _chk(lib().vx_stark_verify_shared(ctypes.cast(self.desc_ptr, _vp), pi.ctypes.data, buf.ctypes.data if buf.size else None, buf.size,
                                          None if sh is None else sh.ctypes.data, api.ctypes.data))
        return api[:self.desc.num_aux_public_inputs].copy()

    def proof_trace_cap(self, proof: bytes) -> np.ndarray:
        cap = np.zeros((1 << self.desc.cap_height, 4), dtype=np.uint64)
        buf = np.frombuffer(proof, dtype=np.uint8)
        _chk(lib().vx_stark_proof_trace_cap(ctypes.cast(self.desc_ptr, _vp), buf.ctypes.data, buf.size, cap.ctypes.data))
        return cap


class AuxDesc(ctypes.Structure):
    """ctypes mirror of `vx_aux_desc` (include/vxprover.h)"""
    _fields_ = [("num_columns", ctypes.c_int32), ("num_challenges", ctypes.c_int32), ("num_fractions", ctypes.c_int32), ("num_sums", ctypes.c_int32),
                ("program_len", ctypes.c_int32), ("program", ctypes.c_void_p), ("sum_coeffs", ctypes.c_void_p), ("fraction_out", ctypes.c_void_p),
                ("sum_out", ctypes.c_void_p)]


class AuxProgram:
    """The second-round columns of ONE challenge set as fractions + running sums (`vx_stark_aux_columns`): computed on the GPU from the
    device-resident trace.  `api_sums`: which running sums' closing values are the table's aux public inputs, in order."""

    def __init__(self, num_columns, num_challenges, program, num_fractions, sum_coeffs, fraction_out=None, sum_out=None, api_sums=()):
        self._prog = (ctypes.c_uint64 * len(program))(*program)
        co = np.ascontiguousarray(np.array(sum_coeffs, dtype=np.int8).reshape(len(sum_coeffs), num_fractions))
        self._co = co
        self._fo = None if fraction_out is None else np.ascontiguousarray(fraction_out, dtype=np.int32)
        self._so = None if sum_out is None else np.ascontiguousarray(sum_out, dtype=np.int32)
        self.num_out = num_fractions + co.shape[0]
        self.num_sums = co.shape[0]
        self.api_sums = tuple(api_sums)
        self.desc = AuxDesc(num_columns, num_challenges, num_fractions, co.shape[0], len(program), ctypes.cast(self._prog, ctypes.c_void_p).value,
                            co.ctypes.data if co.size else None, None if self._fo is None else self._fo.ctypes.data,
                            None if self._so is None else self._so.ctypes.data)

    def precompile(self) -> int:
        """`vx_stark_aux_precompile`: compile the program with hiprtc into the cache (no GPU needed) -> 1 compiled now, 0 cached"""
        rc = lib().vx_stark_aux_precompile(ctypes.byref(self.desc))
        if rc < 0:
            _chk(rc)
        return rc

    def run(self, ctx, d_trace: int, degree_bits: int, challenges, d_out: int) -> np.ndarray:
        """-> the closing sums of the running sums (all of them); the columns are written to device memory at `d_out`"""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64)
        closing = np.zeros(max(1, self.num_sums), dtype=np.uint64)
        _chk(lib().vx_stark_aux_columns(ctx._h, ctypes.byref(self.desc), _vp(d_trace), degree_bits, ch.ctypes.data if ch.size else None, _vp(d_out),
                                        closing.ctypes.data))
        return closing[:self.num_sums]


def stark_verify_bus(tables, proofs) -> np.ndarray:
    """`vx_stark_verify_bus`: tables = [(Stark, public_inputs)] in bus order -> closing sums [num_tables][num_aux_public_inputs];
    raises VxError(VX_E_PROOF, reason) when a proof is invalid or a closing sum does not cancel over the tables."""
    k = len(tables)
    pis = [_as_u64(pi) for _, pi in tables]
    bufs = [np.frombuffer(p, dtype=np.uint8) for p in proofs]
    descs = (ctypes.c_void_p * k)(*[ctypes.cast(st.desc_ptr, _vp).value for st, _ in tables])
    pi_p = (ctypes.c_void_p * k)(*[a.ctypes.data for a in pis])
    pr_p = (ctypes.c_void_p * k)(*[b.ctypes.data for b in bufs])
    lens = (ctypes.c_size_t * k)(*[b.size for b in bufs])
    ns = tables[0][0].desc.num_aux_public_inputs
    out = np.zeros((k, max(1, ns)), dtype=np.uint64)
    _chk(lib().vx_stark_verify_bus(ctypes.cast(descs, _vp), ctypes.cast(pi_p, _vp), ctypes.cast(pr_p, _vp), ctypes.cast(lens, _vp), k, out.ctypes.data))
    return out[:, :ns]


def stark_joint_challenges(caps, cap_heights, n: int) -> np.ndarray:
    """`vx_stark_joint_challenges`: the challenges of a cross-table argument, from every table's trace cap (table order matters)."""
    keep = [np.ascontiguousarray(c, dtype=np.uint64) for c in caps]
    arr = (ctypes.c_void_p * len(caps))(*[k.ctypes.data for k in keep])
    hs = (ctypes.c_int32 * len(caps))(*cap_heights)
    out = np.zeros(n, dtype=np.uint64)
    _chk(lib().vx_stark_joint_challenges(ctypes.cast(arr, _vp), ctypes.cast(hs, _vp), len(caps), n, out.ctypes.data))
    return out


def circuit_precompile(desc_ptr) -> tuple:
    """`vx_circuit_precompile`: compile the constraint-program gates of a circuit with hiprtc ahead of time (no GPU needed; with
    VX_JIT_CACHE_DIR the code objects persist) -> (compiled now, program gates)"""
    n = _i(0)
    rc = lib().vx_circuit_precompile(ctypes.cast(desc_ptr, _vp), ctypes.byref(n))
    if rc < 0:
        _chk(rc)
    return rc, n.value


def circuit_serialize(desc_ptr, constants_sigmas_cap=None, with_preprocessed=True) -> bytes:
    """`vx_circuit_serialize`: a circuit description (+ cap = verifier data, + preprocessed values = prover data) as the
    bytes of a `.vxcircuit` file (csrc/circuit_io.h) — the `build` side of the reference's CLI contract."""
    L = lib()
    cap = None if constants_sigmas_cap is None else _as_u64(constants_sigmas_cap)
    need = L.vx_circuit_serialized_size(ctypes.cast(desc_ptr, _vp), int(cap is not None), int(with_preprocessed))
    if need == 0:
        raise VxError(VX_E_INVALID, L.vx_last_error().decode(errors="replace"))
    buf = np.empty(need, dtype=np.uint8)
    n = _sz(need)
    _chk(L.vx_circuit_serialize(ctypes.cast(desc_ptr, _vp), None if cap is None else cap.ctypes.data, int(with_preprocessed),
                                buf.ctypes.data, ctypes.byref(n)))
    return buf[:n.value].tobytes()


class ParsedCircuit:
    """`vx_circuit_parse`: a description read back from `.vxcircuit` bytes (host only).  `.desc_ptr` can be handed to
    Circuit(), verify_standalone() or the oracle; `.cap` is the stored constants_sigmas cap or None."""

    def __init__(self, data: bytes):
        self._buf = np.frombuffer(data, dtype=np.uint8).copy()      # 8-byte aligned, kept alive: the description borrows it
        d, cap = _vp(), _vp()
        _chk(lib().vx_circuit_parse(self._buf.ctypes.data, self._buf.size, ctypes.byref(d), ctypes.byref(cap)))
        self.desc_ptr = d
        from .synth import CircuitDesc
        self.desc = ctypes.cast(d, ctypes.POINTER(CircuitDesc)).contents
        self.cap = None
        if cap.value:
            n = 1 << self.desc.cap_height
            self.cap = np.frombuffer((ctypes.c_uint64 * (4 * n)).from_address(cap.value), dtype=np.uint64).reshape(n, 4).copy()

    def free(self):
        if self.desc_ptr:
            lib().vx_circuit_desc_free(self.desc_ptr)
            self.desc_ptr = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Circuit:
    """Device-resident prover key — `vx_circuit` (plonky2 `CircuitData`: common + prover-only parts).

    `desc_ptr` is a pointer to a `vx_circuit_desc` (e.g. `vectorx_amd.synth.SynthCircuit.desc_ptr`).
    `prove` mirrors `circuit.prove(&input)` of the reference (/root/reference/circuits/header_range.rs:167)
    at the level of plonky2's `prove_with_partition_witness`: finished witness in, proof bytes out.
    """

    def __init__(self, ctx: Context, desc_ptr, _handle=None, _shape=None):
        self.ctx = ctx
        self._h = _vp()
        if _handle is not None:
            self._h = _handle
            self.degree_bits, self.num_wires, self.cap_height = _shape
        else:
            _chk(lib().vx_circuit_create(ctx._h, ctypes.cast(desc_ptr, _vp), ctypes.byref(self._h)))
            d = ctypes.cast(desc_ptr, ctypes.POINTER(ctypes.c_int32))
            self.degree_bits, self.num_wires, self.cap_height = int(d[0]), int(d[1]), int(d[5])
        ctx._adopt(self)

    @classmethod
    def load(cls, ctx: Context, data: bytes) -> "Circuit":
        """`vx_circuit_load`: `.vxcircuit` bytes -> a device-resident prover key (the `prove` side of the CLI contract)."""
        buf = np.frombuffer(data, dtype=np.uint8)
        hdr = np.frombuffer(data[16:16 + 24], dtype="<i4")
        h = _vp()
        _chk(lib().vx_circuit_load(ctx._h, buf.ctypes.data, buf.size, ctypes.byref(h)))
        return cls(ctx, None, _handle=h, _shape=(int(hdr[0]), int(hdr[1]), int(hdr[5])))

    def free(self):
        if self._h:
            lib().vx_circuit_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def digest(self) -> np.ndarray:
        out = np.empty(4, dtype=np.uint64)
        _chk(lib().vx_circuit_digest(self._h, out.ctypes.data))
        return out

    def constants_sigmas_cap(self) -> np.ndarray:
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        _chk(lib().vx_circuit_constants_sigmas_cap(self._h, out.ctypes.data))
        return out

    def prove(self, wires=None, pow_witness=None, dev_ptr=None) -> bytes:
        """wires: [num_wires][n] column-major host matrix, or dev_ptr: device pointer to the same."""
        cap = lib().vx_proof_size_bound(self._h)
        buf = np.empty(cap, dtype=np.uint8)
        ln = _sz(cap)
        hint = np.array([pow_witness], dtype=np.uint64) if pow_witness is not None else None
        if dev_ptr is None:
            w = _as_u64(wires)
            if w.shape != (self.num_wires, 1 << self.degree_bits):
                raise VxError(VX_E_INVALID, f"witness matrix has shape {w.shape}")
            src, on_dev = w.ctypes.data, 0
        else:
            src, on_dev = dev_ptr, 1
        _chk(lib().vx_prove(self.ctx._h, self._h, src, on_dev, hint.ctypes.data if hint is not None else None,
                            buf.ctypes.data, ctypes.byref(ln)))
        return bytes(buf[:ln.value])

    def verify(self, proof: bytes) -> None:
        """`vx_verify` (CircuitData::verify): raises VxError(VX_E_PROOF, reason) when the proof is not valid."""
        buf = np.frombuffer(proof, dtype=np.uint8) if len(proof) else np.zeros(1, np.uint8)
        _chk(lib().vx_verify(self._h, buf.ctypes.data, len(proof)))

    def program_gates(self):
        """(number of VX_GATE_PROGRAM gates, how many were compiled to native code, why the others were not)"""
        total, compiled = _i(), _i()
        note = ctypes.create_string_buffer(1024)
        _chk(lib().vx_circuit_program_gates(self._h, ctypes.byref(total), ctypes.byref(compiled), note, 1024))
        return total.value, compiled.value, note.value.decode(errors="replace")

    def prove_sharded(self, wires, rank: int, world: int, allgather, user=None, pow_witness=None, dev_ptr=None) -> bytes:
        """`vx_prove_sharded`: this rank's part of ONE proof split across `world` GPUs by LDE coset; every rank
        returns the full proof, byte-identical to `prove`.  `allgather` is the in-place all-gather the host supplies:
        a Python callable `(dev_ptr, bytes_per_rank) -> None` (see vectorx_amd.sharded.TorchAllGather), or a C function
        pointer such as `lib().vx_group_allgather` with `user` = the group member handle."""
        cap = lib().vx_proof_size_bound(self._h)
        buf = np.empty(cap, dtype=np.uint8)
        ln = _sz(cap)
        hint = np.array([pow_witness], dtype=np.uint64) if pow_witness is not None else None
        if dev_ptr is None:
            w = _as_u64(wires)
            if w.shape != (self.num_wires, 1 << self.degree_bits):
                raise VxError(VX_E_INVALID, f"witness matrix has shape {w.shape}")
            src, on_dev = w.ctypes.data, 0
        else:
            src, on_dev = dev_ptr, 1
        err = []
        if callable(allgather) and not isinstance(allgather, ctypes._CFuncPtr):
            def _cb(_user, dptr, nbytes):
                try:
                    allgather(dptr, nbytes)
                    return 0
                except BaseException as e:   # never unwind through the C frames
                    err.append(e)
                    return 1
            fn = ALLGATHER_FN(_cb)
        else:
            fn = allgather
        fn_ptr = ctypes.cast(fn, ctypes.c_void_p) if fn is not None else None
        rc = lib().vx_prove_sharded(self.ctx._h, self._h, src, on_dev, rank, world, fn_ptr, user,
                                    hint.ctypes.data if hint is not None else None, buf.ctypes.data, ctypes.byref(ln))
        if err:
            raise err[0]
        _chk(rc)
        return bytes(buf[:ln.value])

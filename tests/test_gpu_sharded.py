"""ONE proof split across several GPUs by LDE coset (vx_prove_sharded; BASELINE.json configs[3]).

The bar is the same as everywhere: byte-identical proofs.  The test boxes have a single GPU, so the G ranks all use
device 0 — as G host threads exchanging through the library's vx_group (peer copies), and as G processes exchanging
through torch.distributed (gloo, host-staged; backend nccl on a real node)."""
import ctypes
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
import vectorx_amd as vx
from vectorx_amd import sharded
from vectorx_amd.synth import SynthCircuit

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _rank_circuits(sc, world):
    ctxs = [vx.Context(0) for _ in range(world)]
    return ctxs, [vx.Circuit(c, sc.desc_ptr) for c in ctxs]


def _free(ctxs, circuits):
    for c in circuits:
        c.free()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("degree_bits,world", [(12, 4), (14, 8)])
def test_sharded_proof_with_the_recursion_mix_is_byte_identical(oracle, degree_bits, world):
    """the DAG's circuit shape (the recursive verifier's gate set in its declared row mix: the fused gate kernel, the product-tree lookup
    kernel, lookup polynomials from the device) proven over G ranks by coset: every rank's bytes are the oracle's"""
    from vectorx_amd.mapreduce import circuit_shape
    sc = SynthCircuit(degree_bits, seed=950 + degree_bits, witness_seed=5, **circuit_shape(True))
    sc.desc.pow_bits = 8
    w = sc.witness()
    expect = oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    ctxs, circuits = _rank_circuits(sc, world)
    try:
        assert circuits[0].program_gates()[:2] == (9, 9)
        for r, p in enumerate(sharded.prove_sharded_threads(circuits, w)):
            assert p == expect, f"rank {r} of {world}"
    finally:
        _free(ctxs, circuits)


@pytest.mark.parametrize("degree_bits,world,flags", [(3, 2, 0), (4, 8, 0), (5, 4, 0), (6, 2, 0), (6, 8, 7), (8, 4, 1), (10, 8, 0),
                                                     (11, 2, 7), (13, 4, 0), (13, 8, 0), (9, 4, 15)])
def test_sharded_proof_is_byte_identical(oracle, degree_bits, world, flags):
    sc = SynthCircuit(degree_bits, seed=900 + degree_bits, poseidon_percent=40, flags=flags)
    sc.desc.pow_bits = 8
    w = sc.witness()
    expect = oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    ctxs, circuits = _rank_circuits(sc, world)
    try:
        proofs = sharded.prove_sharded_threads(circuits, w)
        assert len(proofs) == world
        for r, p in enumerate(proofs):
            assert p == expect, f"rank {r} of {world}"
    finally:
        _free(ctxs, circuits)


def test_sharded_proof_full_size_verifies(oracle):
    """n = 2^16, 8 ranks: identical to the unsharded GPU proof, accepted by the oracle's verifier."""
    sc = SynthCircuit(16, seed=5, poseidon_percent=50)
    w = sc.witness()
    ctx = vx.Context(0)
    single_c = vx.Circuit(ctx, sc.desc_ptr)
    single = single_c.prove(w)
    ctxs, circuits = _rank_circuits(sc, 8)
    try:
        proofs = sharded.prove_sharded_threads(circuits, w)
        assert all(p == single for p in proofs)
        ov = oracle_lib.OracleCircuit(oracle, sc.desc_ptr, verifier_cap=single_c.constants_sigmas_cap())
        assert ov.verify(proofs[3]) == ""
    finally:
        _free(ctxs, circuits)
        single_c.free()
        ctx.close()


def test_sharded_proof_accepts_pow_hint_and_device_witness():
    sc = SynthCircuit(8, seed=31, poseidon_percent=30)
    sc.desc.pow_bits = 6
    w = sc.witness()
    ctxs, circuits = _rank_circuits(sc, 2)
    try:
        ref = circuits[0].prove(w)
        pw = int(np.frombuffer(ref[-40:-32], dtype="<u8")[0])
        assert all(p == ref for p in sharded.prove_sharded_threads(circuits, w, pow_witness=pw))
    finally:
        _free(ctxs, circuits)


def test_unsatisfied_witness_behaves_like_the_unsharded_prover(oracle):
    """same outcome on every rank as vx_prove — a VX_E_PROOF error or a proof the verifier rejects — and no deadlock"""
    sc = SynthCircuit(7, seed=12, poseidon_percent=50)
    sc.desc.pow_bits = 4
    w = sc.witness().copy()
    w[3, 9] = (int(w[3, 9]) + 1) % oracle_lib.P
    ctxs, circuits = _rank_circuits(sc, 4)
    try:
        try:
            single = circuits[0].prove(w)
        except vx.VxError as e:
            single = e.code
        try:
            proofs = sharded.prove_sharded_threads(circuits, w)
        except vx.VxError as e:
            assert single == e.code == vx.VX_E_PROOF
        else:
            assert all(p == single for p in proofs)
            assert oracle_lib.OracleCircuit(oracle, sc.desc_ptr).verify(proofs[0]) != ""
    finally:
        _free(ctxs, circuits)


def test_a_rank_that_dies_mid_proof_does_not_hang_its_peers():
    """rank 2 of 4 stops taking part after its third exchange WITHOUT aborting the group (a crashed host
    thread cannot).  The peers must come back with VX_E_COMM — through the barrier timeout — not wait forever; and a rank
    that fails with an error between exchanges aborts the group so that the peers return at once."""
    import threading
    import time
    sc = SynthCircuit(8, seed=77, poseidon_percent=40)
    sc.desc.pow_bits = 4
    w = sc.witness()
    world = 4
    L = vx.lib()
    for mode in ("dies", "aborts"):
        ctxs, circuits = _rank_circuits(sc, world)
        g = ctypes.c_void_p()
        assert L.vx_group_create(world, ctypes.byref(g)) == 0
        assert L.vx_group_set_timeout_ms(g, 3000) == 0
        members = []
        for r in range(world):
            m = ctypes.c_void_p()
            assert L.vx_group_join(g, r, ctxs[r]._h, ctypes.byref(m)) == 0
            members.append(m)
        calls = {"n": 0}

        def flaky(dptr, nbytes):                       # rank 2's exchange: three good ones, then the rank is gone
            calls["n"] += 1
            if calls["n"] > 3:
                if mode == "aborts":
                    L.vx_group_abort(g)
                raise RuntimeError("rank 2 died")
            rc = L.vx_group_allgather(members[2], ctypes.c_void_p(dptr), nbytes)
            if rc != 0:
                raise vx.VxError(rc, "exchange failed")

        out = [None] * world

        def run(r):
            try:
                if r == 2:
                    circuits[r].prove_sharded(w, r, world, flaky)
                else:
                    circuits[r].prove_sharded(w, r, world, L.vx_group_allgather, members[r])
                out[r] = "proof"
            except BaseException as e:
                out[r] = e

        t0 = time.time()
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
        took = time.time() - t0
        assert not any(t.is_alive() for t in threads), "a rank is still waiting for the dead one"
        assert isinstance(out[2], RuntimeError)
        for r in (0, 1, 3):
            assert isinstance(out[r], vx.VxError) and out[r].code == vx.VX_E_COMM, (mode, r, out[r])
        assert took < (20 if mode == "dies" else 10), took
        L.vx_group_destroy(g)
        _free(ctxs, circuits)


def test_sharded_argument_checks(ctx):
    sc = SynthCircuit(5, seed=1, poseidon_percent=50)
    c = vx.Circuit(ctx, sc.desc_ptr)
    w = sc.witness()
    noop = lambda ptr, nbytes: None  # noqa: E731
    for rank, world in ((0, 3), (0, 16), (2, 2), (-1, 2), (0, 0)):
        with pytest.raises(vx.VxError) as e:
            c.prove_sharded(w, rank, world, noop)
        assert e.value.code == vx.VX_E_INVALID
    with pytest.raises(vx.VxError):
        c.prove_sharded(w, 0, 2, None)                      # world > 1 without a callback
    assert c.prove_sharded(w, 0, 1, None) == c.prove(w)      # world 1 needs none

    def boom(ptr, nbytes):
        raise RuntimeError("link down")
    with pytest.raises(RuntimeError, match="link down"):     # a failing exchange surfaces, the library returns VX_E_COMM
        c.prove_sharded(w, 0, 2, boom)
    assert c.prove(w) == c.prove_sharded(w, 0, 1, None)      # and the context stays usable
    c.free()


def test_group_abort_wakes_waiting_ranks(ctx):
    L = vx.lib()
    g = ctypes.c_void_p()
    assert L.vx_group_create(2, ctypes.byref(g)) == 0
    m = ctypes.c_void_p()
    assert L.vx_group_join(g, 0, ctx._h, ctypes.byref(m)) == 0
    import threading
    rc = []
    buf = ctypes.c_void_p()
    assert L.vx_dev_alloc(ctx._h, 64, ctypes.byref(buf)) == 0
    t = threading.Thread(target=lambda: rc.append(L.vx_group_allgather(m, buf, 32)))
    t.start()
    t.join(0.3)
    assert t.is_alive()                                      # rank 1 never arrives: rank 0 is waiting
    L.vx_group_abort(g)
    t.join(10)
    assert not t.is_alive() and rc == [vx.VX_E_COMM]
    L.vx_dev_free(ctx._h, buf)
    L.vx_group_destroy(g)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,degree_bits,flags", [(2, 9, 0), (4, 7, 5)])
def test_sharded_proof_across_processes_torch_distributed(world, degree_bits, flags):
    env = dict(os.environ, OMP_NUM_THREADS="1", VX_TEST_DEGREE_BITS=str(degree_bits), VX_TEST_FLAGS=str(flags),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "_mp_sharded_prove_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["world"] == world
    res = sorted(out["results"])
    assert [x[0] for x in res] == list(range(world))
    assert all(x[1] for x in res)                            # every rank's proof == its own unsharded proof
    assert len({x[2] for x in res}) == 1                     # and all ranks hold the same bytes
    assert all(x[3] >= 8 for x in res)                       # witness, caps x3, quotient coefficients, FRI cap + layer, openings


def test_torch_allgather_on_device_memory_nccl_world1():
    """The production exchange path (backend nccl = RCCL): torch views the library's device buffer in place.  With one
    GPU only world 1 can run here; it still proves that the zero-copy view of a vx_dev_alloc pointer works."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VX_FORCE_DIST="1", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "import vectorx_amd as vx\n"
        "from vectorx_amd import sharded\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "ctx = vx.Context(0)\n"
        "p = ctx.alloc(4096)\n"
        "a = np.arange(512, dtype=np.uint64) * 7\n"
        "ctx.upload(p, a)\n"
        "ag = sharded.TorchAllGather(ctx, dist, torch.device('cuda', 0))\n"
        "ag(p, 4096)\n"
        "assert (ctx.download(p, 4096) == a).all()\n"
        "v = torch.as_tensor(sharded._DeviceView(p, 4096), device=torch.device('cuda', 0))\n"
        "v[:8] = 0\n"
        "torch.cuda.synchronize()\n"
        "assert int(ctx.download(p, 4096)[0]) == 0 and int(ctx.download(p, 4096)[1]) == 7\n"
        "ctx.free(p); ctx.close(); dist.destroy_process_group(); print('OK')\n"
    ) % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# ---- real multi-GPU paths: skipped on the single-GPU test boxes, exercised wherever >= 2 devices are visible --------
def _need_devices(k):
    n = vx.lib().vx_device_count()
    if n < k:
        pytest.skip(f"needs {k} GPUs, {n} visible (single-device boxes run the same code with every rank on device 0)")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_proof_one_context_per_device(oracle, world):
    """The G ranks on G DISTINCT devices: vx_group_join enables peer access pairwise, the all-gathers are
    hipMemcpyPeerAsync over xGMI, and every rank still returns the oracle's proof byte for byte."""
    _need_devices(world)
    sc = SynthCircuit(12, seed=4242, poseidon_percent=50)
    sc.desc.pow_bits = 8
    w = sc.witness()
    expect = oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    ctxs = [vx.Context(r) for r in range(world)]
    circuits = [vx.Circuit(c, sc.desc_ptr) for c in ctxs]
    try:
        for p in sharded.prove_sharded_threads(circuits, w):
            assert p == expect
    finally:
        _free(ctxs, circuits)


@pytest.mark.parametrize("mode", ["throughput", "sharded"])
def test_bench_two_gpus_under_torchrun(mode):
    """bench.py --gpus 2 as the driver launches it (one process per GPU, RCCL): the launcher starts before any GPU call."""
    _need_devices(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "16",
           "--mode", mode, "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["scaling"] == ("strong" if mode == "sharded" else "weak")


def _plain_bench(*extra):
    """`python bench.py --gpus N ...` run PLAINLY — no launcher in front: bench.py starts its own ranks as child processes"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # --full-line: the complete line (whole digests, the descriptions) instead of the compact form the driver's record keeps
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--log-n", "16", "--no-cpu-baseline", "--full-line", *extra],
                          capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)


@pytest.mark.parametrize("mode", ["throughput", "sharded"])
def test_bench_gpus_2_run_plainly_launches_two_ranks(mode):
    """the driver's command form is `python3 bench.py --gpus N ...`: it must BE an N-rank run"""
    _need_devices(2)
    r = _plain_bench("--gpus", "2", "--mode", mode, "--dag-spec", "8,14,12,15", "--dag-starks-small")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["value"] > 0
    assert len({d["local_rank"] for d in line["rank_devices"]}) == 2
    if mode == "sharded":
        # bench.py hands vx_prove_sharded a DEVICE-resident witness: no witness all-gather, so 8 exchanges per proof (3 + 1 caps, quotient
        # coset coefficients, the openings, first FRI layer, ONE batched query-opening exchange); a host witness adds the column all-gather = 9
        assert line["exchange"]["allgather_calls_per_proof"] == EXCHANGES_PER_PROOF["device"]
        assert line["exchange"]["inbound_bytes_per_rank_per_proof"] > 0
    _check_multi_rank_legs(line, 2, "nccl")


# exchanges vx_prove_sharded asks its host for, by where the witness is (prover.hip.h: the witness-column all-gather exists only for a host witness)
# round 6: + the openings (the ~260 evaluations at zeta / g zeta shard by column range, 2 words per polynomial come back)
EXCHANGES_PER_PROOF = {"device": 8, "host": 9}


def _check_multi_rank_legs(line, world, backend):
    """the two legs bench.py adds after the timed region of an N > 1 run"""
    sh = line["sharded_one_proof"]
    assert "error" not in sh, sh
    assert sh["ranks"] == world and sh["ms_per_proof"] > 0 and sh["backend"] == backend
    assert sh["allgather_calls_per_proof"] == EXCHANGES_PER_PROOF["device"] and sh["inbound_bytes_per_rank_per_proof"] > 0
    assert sh["all_ranks_returned_the_same_proof"] is True and sh["byte_identical_to_unsharded_vx_prove"] is True
    assert sh["identical_to_unsharded"] is True and sh["rccl_ranks"] == world and sh["exchange_host_wait_ms"] >= 0
    assert "exchange_host_wait_ms_rank0" in sh
    # the line names every rank's device: N ranks, each with a PCI bus id (N DISTINCT ids on a real node, one on the emulation)
    assert line["rccl_ranks"] == world and len(line["rank_devices"]) == world
    assert all(isinstance(d["pci_bus_id"], int) and d["pci_bus_id"] >= 0 for d in line["rank_devices"])
    dag = line["dag_header_range_512"]
    assert "error" not in dag, dag
    assert dag["ranks"] == world and dag["dag_seconds"] > 0 and len(dag["root"]) == 64
    assert dag["plonky2_proofs"] == sum(l[1] for l in dag["per_layer_ms"])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_multi_rank_code_path_on_one_device(world):
    """`python bench.py --gpus N` exactly as the driver runs it, but with the N ranks as processes on DEVICE 0 over gloo (RCCL refuses
    duplicate devices): the weak-scaling line + the sharded-one-proof leg + the N-rank DAG leg, tiny sizes.  The single-GPU boxes
    run this every round, so the code the 8-GPU driver run takes has been executed before it gets there."""
    r = _plain_bench("--gpus", str(world), "--ranks-on-one-device", "--log-n", "12" if world < 8 else "10", "--dag-spec", "4,10,9,11" if world < 8 else "8,9,8,10",
                     "--sharded-leg-steps", "2", "--dag-workers", "1", "--dag-lanes", "2",
                     *(["--dag-starks-small"] if world == 2 else ["--no-dag-stark-leg"]), *(["--no-dag-pool-leg"] if world == 8 else []))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["rccl_ranks"] == world and line["dist_backend"] == "gloo" and line["value"] > 0
    assert line["scaling"] == "weak" and "emulated_ranks_on_one_device" in line
    import bench_prove
    short = bench_prove.compact_line(line)                       # what the same run prints without --full-line
    assert len(json.dumps(short)) < 6144 and short["value"] > 0 and short["emulated_ranks_on_one_device"] is True
    assert short["sharded_one_proof"]["allgather_calls_per_proof"] == EXCHANGES_PER_PROOF["device"] and short["dag_header_range_512"]["dag_seconds"] > 0
    assert [d["rank"] for d in line["rank_devices"]] == list(range(world))
    _check_multi_rank_legs(line, world, "gloo")
    assert line["dag_header_range_512"]["plonky2_proofs"] == (4 + 3 + 1 if world < 8 else 8 + 7 + 1)
    if world < 8:       # the same DAG once more on ONE pool of worker processes over all "GPUs" (here: one worker per rank's device, all on device 0)
        pool = line["dag_on_one_pool_over_all_gpus"]
        assert "error" not in pool, pool
        assert pool["devices"] == [0] * world
        p0 = pool["dag_header_range_512"]
        assert p0["plonky2_proofs"] == 8 and len(p0["jobs_by_worker"]) == world and len(p0["root"]) == 64
        if world == 2:
            assert pool["dag_header_range_512_with_starks"]["lane_seconds_by_kind"]["trace_generation"] > 0
            assert pool["dag_header_range_512_with_starks"]["output_equals_host_computation"] is True
    if world == 2:      # the same DAG with every job's STARK tables (smallest shapes), over both ranks
        ds = line["dag_header_range_512_with_starks"]
        assert "error" not in ds, ds
        assert ds["with_stark_tables"] is True and ds["ranks"] == 2 and ds["dag_seconds"] > 0
        assert set(ds["rank0_lane_seconds_by_kind"]) >= {"plonky2", "blake2b", "sha256", "signature_bus", "trace_generation"}
        assert ds["root"] != line["dag_header_range_512"]["root"]          # the STARK proofs are part of every job's digest
        assert ds["output_equals_host_computation"] is True and len(ds["output"]) == 192          # statements cross the ranks with the digests


def test_bench_line_survives_multi_rank_legs_that_never_finish():
    """a collective leg that hangs must not cost the contract's line: at the deadline every rank ends cleanly, rank 0 having printed the
    line without the legs (here the deadline is already over when the legs start)"""
    r = _plain_bench("--gpus", "2", "--ranks-on-one-device", "--log-n", "12", "--dag-spec", "4,10,9,11", "--multi-rank-leg-deadline", "0.001")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "did not finish" in line["extra_legs"]["error"]


def test_bench_refuses_more_gpus_than_visible():
    """asking for more GPUs than the box has must fail loudly — never a smaller run that prints a result line"""
    n = vx.lib().vx_device_count()
    r = _plain_bench("--gpus", str(n + 1))
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


def test_bench_single_gpu_line_names_its_device():
    r = _plain_bench("--gpus", "1", "--no-host-witness-leg")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and len(line["rank_devices"]) == 1


@pytest.mark.parametrize("degree_bits,world,flags", [(6, 2, 32), (9, 4, 32 | 7), (11, 8, 32 | 16)])
def test_sharded_proof_with_u32_gates(oracle, degree_bits, world, flags):
    """the U32 / comparison program gates under the coset split: every rank evaluates them on its own cosets"""
    sc = SynthCircuit(degree_bits, seed=3300 + degree_bits, poseidon_percent=40, flags=flags)
    sc.desc.pow_bits = 6
    w = sc.witness()
    expect = oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    ctxs, circuits = _rank_circuits(sc, world)
    try:
        for p in sharded.prove_sharded_threads(circuits, w):
            assert p == expect
    finally:
        _free(ctxs, circuits)


@pytest.mark.parametrize("degree_bits,world,flags", [(7, 2, 16), (9, 4, 16 | 1), (11, 8, 16)])
def test_sharded_proof_with_lookup_argument(oracle, degree_bits, world, flags):
    sc = SynthCircuit(degree_bits, seed=1900 + degree_bits, poseidon_percent=40, flags=flags)
    sc.desc.pow_bits = 6
    w = sc.witness()
    expect = oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    ctxs, circuits = _rank_circuits(sc, world)
    try:
        for p in sharded.prove_sharded_threads(circuits, w):
            assert p == expect
    finally:
        _free(ctxs, circuits)


@pytest.mark.parametrize("degree_bits,world,flags,qdf", [(7, 2, 1, 4), (9, 8, 16, 5), (10, 4, 16 | 1, 7), (8, 4, 0, 3)])
def test_sharded_proof_with_a_quotient_degree_factor_below_the_blowup(oracle, degree_bits, world, flags, qdf):
    """every rank evaluates its cosets of the full 8n domain; the chunk transform trims to qdf chunks per challenge"""
    sc = SynthCircuit(degree_bits, seed=2900 + degree_bits, poseidon_percent=40, flags=flags, quotient_degree_factor=qdf)
    sc.desc.pow_bits = 6
    w = sc.witness()
    expect = oracle_lib.OracleCircuit(oracle, sc.desc_ptr).prove(w)
    ctxs, circuits = _rank_circuits(sc, world)
    try:
        for p in sharded.prove_sharded_threads(circuits, w):
            assert p == expect
    finally:
        _free(ctxs, circuits)


# ---- STARKs sharded by coset (vx_stark_begin_sharded) -------------------------------------------------------------
def _stark_case(name, degree_bits, rate_bits):
    import stark_airs as airs
    cfg = dict(rate_bits=rate_bits, num_query_rounds=10, pow_bits=4)
    if name == "sha256":
        from vectorx_amd import sha256_air as sha
        stark = sha.make_stark(degree_bits, **cfg)
        trace, pis, _ = sha.generate_trace(degree_bits, [b"abc", b"", bytes(range(100))])
        return stark, trace, pis
    return getattr(airs, name)(degree_bits, **cfg)


@pytest.mark.parametrize("name,degree_bits,rate_bits,world", [
    ("mulchain", 10, 1, 2), ("logup", 9, 1, 2), ("sha256", 9, 1, 2),              # starky's rate_bits = 1: two ranks, one coset each
    ("fibonacci", 8, 3, 8), ("fibonacci", 8, 2, 2), ("cubic", 8, 3, 4), ("cubic", 9, 2, 4), ("mulchain", 10, 3, 8), ("mulchain", 9, 3, 2),
    ("logup", 9, 3, 4), ("logup", 8, 3, 8), ("sha256", 8, 2, 4)])
def test_stark_proof_sharded_by_coset_is_byte_identical(ctx, name, degree_bits, rate_bits, world):
    """the G ranks as host threads with one context each ON ONE DEVICE (single-device emulation, like the plonk path's tests): every
    rank's proof equals the unsharded vx_stark_prove / begin + finish proof byte for byte — quotient domains of 1 (fibonacci), 2
    (degree-3 AIRs) blocks, with fewer, as many and more ranks than quotient blocks, with and without a second commitment round"""
    stark, trace, pis = _stark_case(name, degree_bits, rate_bits)
    expect = stark.prove(ctx, trace, pis)
    ctxs = [vx.Context(0) for _ in range(world)]
    try:
        proofs = sharded.prove_stark_sharded_threads(ctxs, stark, trace, pis)
        assert all(p == expect for p in proofs)
    finally:
        for c in ctxs:
            c.close()
    stark.verify(pis, expect) if stark.desc.num_aux_public_inputs == 0 else None


def test_stark_sharded_refuses_bad_worlds(ctx):
    import ctypes
    import stark_airs as airs
    stark, trace, pis = airs.mulchain(8, rate_bits=1, num_query_rounds=10, pow_bits=4)
    L = vx.lib()
    sess = ctypes.c_void_p()
    t = np.ascontiguousarray(trace, dtype=np.uint64)
    for rank, world in ((0, 4), (0, 3), (2, 2), (-1, 2)):       # 4 > 2^rate_bits, not a power of two, rank outside the world
        rc = L.vx_stark_begin_sharded(ctx._h, ctypes.cast(stark.desc_ptr, ctypes.c_void_p), t.ctypes.data, 0, pis.ctypes.data, rank, world,
                                      ctypes.cast(L.vx_group_allgather, ctypes.c_void_p), None, None, ctypes.byref(sess))
        assert rc == vx.VX_E_INVALID, (rank, world)
    rc = L.vx_stark_begin_sharded(ctx._h, ctypes.cast(stark.desc_ptr, ctypes.c_void_p), t.ctypes.data, 0, pis.ctypes.data, 0, 2, None, None, None,
                                  ctypes.byref(sess))
    assert rc == vx.VX_E_INVALID                                # two ranks need an all-gather

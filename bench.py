#!/usr/bin/env python3
"""bench.py — throughput of the VectorX prover hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path — one whole plonky2 proof (`vx_prove`) — over one synthetic
header_range_512-shaped witness (BASELINE.json configs[2]: n = 2^21 rows x 135 wire columns, blow-up 8,
cap height 4, full FRI) that is already resident in HBM when the timed region starts.  With N GPUs every
rank proves its own witness (proof-level sharding — the reference's MapReduce fan-out,
/root/reference/circuits/builder/subchain_verification.rs:72-78 — no data-path collective), so scaling is
"weak" and `value` is the whole-job rate.  One JSON line is printed by rank 0; it carries `roofline`
(HIP-event-timed coset-LDE NTT kernel family on the library's own stream) and `cpu_baseline` (the oracle =
CPU restatement of plonky2 v0.2.0, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=21, help="log2 trace rows (21 = header_range_512 stand-in)")
    ap.add_argument("--ncols", type=int, default=135)
    ap.add_argument("--workload", default="prove", choices=["commit", "prove"])
    ap.add_argument("--mode", default="throughput", choices=["throughput", "sharded"],
                    help="N>1 only. throughput (default, BASELINE configs[4]): one proof per GPU, weak scaling.  sharded "
                         "(configs[3]): ALL GPUs work on one proof (vx_prove_sharded, coset split, RCCL all-gathers), strong scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-n", type=int, default=None,
                    help="rows (log2) of the bounded CPU-baseline sample (default 18 for prove: ~20 s on 16 cores; 17 for commit)")
    ap.add_argument("--poseidon-percent", type=int, default=50, help="share of PoseidonGate rows in the synthetic circuit")
    ap.add_argument("--no-dag-leg", action="store_true",
                    help="skip the extra (untimed-by-the-contract) leg that proves one whole header_range_512 DAG: 64 map + 63 reduce + 1 outer proofs")
    ap.add_argument("--multi-rank-leg-deadline", type=float, default=1500.0,
                    help="seconds the world > 1 legs may take together before every rank gives up on them and rank 0 prints the line without them")
    ap.add_argument("--no-dag-stark-leg", action="store_true",
                    help="skip the extra leg that proves the header_range_512 DAG WITH the STARK tables of every job (BLAKE2b / SHA-256 / SHA-512 / batched EdDSA)")
    ap.add_argument("--dag-workers", type=int, default=3,
                    help="worker PROCESSES per GPU of the DAG legs (vectorx_amd/dag_pool.py; each keeps --dag-lanes jobs in flight); 0 = the one-process scheduler of rounds 1-4")
    ap.add_argument("--dag-lanes", type=int, default=3,
                    help="jobs in flight per worker process (3 workers x 3 lanes: 3.38 s against 3.50 for 2 x 3 and 3 x 2, 3.42 for 4 x 2 on one box — profiles/r05_dag_pool.jsonl; 3 x 4 and 4 x 3 are no faster)")
    ap.add_argument("--no-rotate-leg", action="store_true", help="skip the rotate leg (one rotate request: plonky2 2^19 + its tables; N = 1 only)")
    ap.add_argument("--no-dag-pool-leg", action="store_true",
                    help="N > 1 only: skip the leg that runs the DAG on ONE pool of worker processes spanning all N GPUs (rank 0 coordinates; vectorx_amd/dag_pool.py)")
    ap.add_argument("--dag-table-mode", default="per_job", choices=["per_job", "resident"],
                    help="STARK tables of the DAG: per_job = every job's own inputs, traces generated on the GPU inside the clock; resident = one host-generated trace per table kind (rounds 3-4)")
    ap.add_argument("--extra-legs-deadline", type=float, default=1500.0,
                    help="seconds the legs AFTER the contract's line may take at N = 1 before the process prints what it has and exits")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="(default since round 6; kept so that older command lines still parse)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the COMPLETE line (descriptions, whole digests: 12-16 KB) instead of the compact form (< 6 KB: every number, no prose; "
                         "profiles/bench_line_glossary.md describes the fields).  The complete line is always written to bench_line_full.json")
    ap.add_argument("--no-cpu-baseline-full", action="store_true",
                    help="cpu_baseline: only the bounded sample (2^18 rows, scaled); by default the N = 1 run ALSO proves the bench circuit itself once "
                         "with the oracle — ~4 min of host cores after every GPU leg — and `cpu_baseline.value` is that measured figure")
    ap.add_argument("--no-chip-leg", action="store_true",
                    help="skip the extra leg that proves the three chip-sized STARK tables (SHA-256, BLAKE2b, Ed25519 scalar multiplication; SURVEY §8 f-3)")
    ap.add_argument("--circuit-flags", type=int, default=0,
                    help="vectorx_amd.synth FLAG_* bits of the synthetic circuit (1|4|8: constraint-program gates, 16: a lookup table); the "
                         "default 0 is the headline gate mix — other values are for profiling the prove-only kernels")
    ap.add_argument("--recursion-mix", action="store_true",
                    help="profiling: the workload circuit carries the recursive verifier's gate set in its declared row mix (the DAG legs' circuit family, "
                         "vectorx_amd/synth.py RECURSIVE_VERIFIER_MIX) instead of the headline two-gate mix; implies no extra legs, like --circuit-flags")
    ap.add_argument("--no-host-witness-leg", action="store_true",
                    help="skip the extra (untimed-by-the-contract) leg that proves from a pinned HOST witness (PCIe-inclusive rate)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 run: nccl (= RCCL over xGMI, the default and the only one a result may be "
                         "quoted from) or gloo (host-staged exchanges; for --ranks-on-one-device)")
    ap.add_argument("--ranks-on-one-device", action="store_true",
                    help="EMULATION for single-GPU boxes: the N ranks are N processes that all use device 0 (RCCL refuses duplicate devices, "
                         "so this forces --dist-backend gloo).  Exercises the N > 1 code path; the line is labelled and is not a measurement")
    ap.add_argument("--no-multi-rank-legs", action="store_true",
                    help="N > 1 only: skip the two extra legs after the timed region (one proof sharded over all ranks = BASELINE configs[3]; "
                         "one header_range_512 DAG over all ranks)")
    ap.add_argument("--sharded-leg-steps", type=int, default=3)
    ap.add_argument("--dag-starks-small", action="store_true",
                    help="N > 1 legs: the STARK tables of the DAG at the smallest shapes they allow (the single-GPU emulation test)")
    ap.add_argument("--dag-spec", default="64,18,16,19",
                    help="num_map,map_log_n,reduce_log_n,outer_log_n of the multi-rank DAG leg (tests pass tiny sizes)")
    return ap.parse_args()


def synth_witness_matrix(seed, ncols, n):
    """uniform field elements from a fixed-seed PRNG; column-major [ncols][n]"""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 0xFFFFFFFF00000001, size=(ncols, n), dtype=np.uint64)


def cpu_baseline_commit(args):
    """oracle (kind = "port"): PolynomialBatch::from_values on a bounded sample, all host cores."""
    import oracle_lib
    import bench_prove
    oracle = oracle_lib.load()
    cores = bench_prove.usable_cores()   # affinity capped by the cgroup CPU quota
    oracle.L.vxo_set_num_threads(cores)
    s_log = min(args.cpu_sample_log_n, args.log_n)
    vals = synth_witness_matrix(1234, args.ncols, 1 << s_log)
    t0 = time.perf_counter()
    oracle.commit(vals, 3, 4, want_leaves=False)
    dt = time.perf_counter() - t0
    scale = float(1 << (args.log_n - s_log))  # rows scale linearly (the log factor favours the CPU slightly)
    return {
        "value": 1.0 / (dt * scale), "unit": "trace commits/sec", "cores": cores, "kind": "port",
        "sample": f"oracle PolynomialBatch::from_values on [{args.ncols}][2^{s_log}] took {dt:.2f} s with {cores} "
                  f"OpenMP threads; scaled x{int(scale)} to 2^{args.log_n} rows",
    }


def _pci_bus_id(torch, idx) -> int:
    p = torch.cuda.get_device_properties(idx)
    return int(getattr(p, "pci_bus_id", -1))


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` run plainly (no launcher): start the N ranks as CHILD processes through
    torch.distributed.run — before this process has made a single GPU call, and never by exec — relay rank 0's JSON
    line and return the launcher's exit code.  Fewer than N visible devices is an error, not a smaller run."""
    import torch
    ndev = torch.cuda.device_count()      # does not initialise the GPU on this image
    if ndev < (1 if args.ranks_on_one_device else args.gpus):
        print(f"bench.py: --gpus {args.gpus} asked for but only {ndev} GPU(s) are visible — refusing to run a smaller job "
              f"under that label", file=sys.stderr)
        return 3
    with socket.socket() as s:            # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["VX_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if r.returncode == 0 and line is None:
        print("bench.py: the ranks exited 0 but rank 0 printed no result line", file=sys.stderr)
        return 4
    if line is None:
        return r.returncode
    if r.returncode != 0:
        # rank 0 prints its line once the timed region and every single-rank leg are complete; only the multi-rank legs and the teardown come
        # after it.  A rank lost there must not cost the measurement: the line goes out, carrying the launcher's exit code.
        d = json.loads(line)
        d["ranks_exit_code_after_the_line"] = r.returncode
        line = json.dumps(d)
        print(f"bench.py: the ranks ended with exit code {r.returncode} AFTER rank 0 had printed its line", file=sys.stderr)
    print(line, flush=True)
    return 0


def main():
    args = parse()
    if args.recursion_mix:
        args.circuit_flags = args.circuit_flags or 29      # (the flag set the mix needs; also switches the extra legs off)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    # The DAG legs' worker processes (vectorx_amd/dag_pool.py) are started HERE, before this process makes its first GPU call (children,
    # never an exec); they connect back and idle — no GPU work, no CPU load — until the legs configure them, after the timed region.
    jit_cache = ROOT / ".jit_cache"
    if "VX_JIT_CACHE_DIR" not in os.environ and jit_cache.is_dir():     # compiled constraint programs (__graft_entry__.build); the workers inherit it
        os.environ["VX_JIT_CACHE_DIR"] = str(jit_cache)
    dag_pool = None
    pool_multi = (args.gpus > 1 and args.workload == "prove" and not args.no_dag_leg and not args.no_multi_rank_legs and not args.no_dag_pool_leg
                  and not args.circuit_flags and args.dag_workers > 0)        # the same on every rank: decided from the arguments alone
    if ("WORLD_SIZE" not in os.environ and args.gpus == 1 and args.workload == "prove" and not args.no_dag_leg and args.log_n >= 20
            and not args.circuit_flags and args.dag_workers > 0):
        pool_devices = (0,)
    elif pool_multi and int(os.environ.get("RANK", "0")) == 0:
        # N > 1: rank 0 coordinates ONE pool whose workers sit on all N GPUs (on device 0, all of them, under --ranks-on-one-device)
        pool_devices = tuple([0] * args.gpus if args.ranks_on_one_device else range(args.gpus))
    else:
        pool_devices = None
    if pool_devices is not None:
        from vectorx_amd import mapreduce as mr
        from vectorx_amd.dag_pool import DagPool
        dag_pool = DagPool(mr.DagSpec(*(int(x) for x in args.dag_spec.split(","))), devices=pool_devices, workers_per_device=args.dag_workers,
                           lanes=args.dag_lanes, with_starks=not args.no_dag_stark_leg, small_tables=args.dag_starks_small,
                           table_mode=args.dag_table_mode).start()
    import torch
    import vectorx_amd as vx
    from vectorx_amd import dist_harness as H

    rank, world, local_rank = H.env_rank()
    if args.ranks_on_one_device:
        args.dist_backend = "gloo"        # RCCL refuses two ranks on one device
        local_rank = 0                    # every rank is a process on device 0: an emulation, labelled as such in the line
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} has LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    dist = H.init(args.dist_backend, local_rank)
    on_host = dist is not None and dist.get_backend() != "nccl"     # gloo: collectives on host tensors
    # a HOST-side barrier group for the leg in which rank 0 alone drives the GPUs (the worker pool): waiting inside an RCCL collective would
    # park a kernel on every GPU the pool's workers are trying to use
    host_group = dist.new_group(backend="gloo") if (dist is not None and pool_multi and not on_host) else None
    ctx = vx.Context(local_rank)  # no CPU fallback: raises if the HIP library / GPU is missing
    n = 1 << args.log_n
    if args.cpu_sample_log_n is None:
        args.cpu_sample_log_n = 18 if args.workload == "prove" else 17
    keep = []

    if args.workload == "commit":
        host = synth_witness_matrix(0x5EED0000 + rank, args.ncols, n)
        d_in = ctx.alloc(host.nbytes)
        ctx.upload(d_in, host)
        del host

        def step():
            b = vx.PolynomialBatch.from_values_dev(ctx, d_in, args.log_n, args.ncols, 3, 4)
            cap = b.cap()
            b.free()
            return cap
        metric = "header_range_512 trace commits/sec (PolynomialBatch::from_values of the 135-column wire trace; stage 1 of prove())"
        unit = "trace commits/sec"
        wl_name = f"header_range_512 stand-in: wires commit, n=2^{args.log_n} rows x {args.ncols} cols, blowup 8, cap_height 4"
        cleanup = lambda: ctx.free(d_in)
    else:
        import bench_prove
        step, metric, unit, wl_name, cleanup = bench_prove.make_step(ctx, args, rank, dist, torch.device("cuda", local_rank))

    def sync():
        ctx.sync()
        torch.cuda.synchronize()

    def before_timed():
        ctx.prof_enable(True)
        ctx.prof_reset()
        if args.workload == "prove" and "allgather" in bench_prove._LEG:   # sharded mode: count the timed steps' exchanges only
            bench_prove._LEG["allgather"].calls = bench_prove._LEG["allgather"].bytes = 0

    dt = H.run_timed(step, args.steps, args.warmup, sync, dist, device=None if on_host else f"cuda:{local_rank}", before_timed=before_timed)
    prof = ctx.prof()
    # which device every rank ran on — gathered over RCCL, so the line proves N distinct GPUs took part
    rank_devices, rccl_ranks = [{"rank": 0, "local_rank": local_rank, "pci_bus_id": _pci_bus_id(torch, local_rank)}], 1
    if dist is not None:
        rccl_ranks = dist.get_world_size()
        mine = torch.tensor([rank, local_rank, _pci_bus_id(torch, local_rank)], dtype=torch.int64, device="cpu" if on_host else f"cuda:{local_rank}")
        slots = [torch.empty_like(mine) for _ in range(rccl_ranks)]
        dist.all_gather(slots, mine)
        rank_devices = [{"rank": int(t[0]), "local_rank": int(t[1]), "pci_bus_id": int(t[2])} for t in (x.cpu() for x in slots)]
    hash_clock_ghz = ctx.clock_ghz() if args.workload == "prove" else None   # before anything else runs: the timed region's own samples
    ctx.prof_enable(False)

    # Still before the line: the same K proofs from a page-locked HOST witness — SURVEY.md §8d's end-to-end definition (witness in
    # host memory -> proof bytes, PCIe included).  bench.py's contract keeps `value` = "inputs resident in HBM when the timed region
    # starts" (the PCIe-inclusive rate is never `value`), so the end-to-end figure rides next to it as `value_end_to_end`.
    host_leg = None
    if args.workload == "prove" and world == 1 and not args.no_host_witness_leg:
        import bench_prove
        host_leg = bench_prove.host_witness_leg(ctx, args, sync)

    # N > 1: two more legs after the timed region so that ONE `bench.py --gpus N` yields the weak-scaling point
    # (`value`), the strong-scaling point (one 2^21 proof over all N ranks = BASELINE configs[3]) and the real unit of work (one
    # header_range_512 DAG over all N ranks: /root/reference/circuits/builder/subchain_verification.rs:72-78).  Collective:
    # every rank takes part; an error on any rank is reported by rank 0 instead of costing the contract's line.
    out = None
    if rank == 0:
        lde = prof.get("lde", {"ms": 0.0, "calls": 0, "alg_bytes": 0.0})
        roof = None
        if lde["calls"]:
            per_launch_ms = lde["ms"] / lde["calls"]
            per_launch_bytes = lde["alg_bytes"] / lde["calls"]
            achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            # HBM bytes per launch: PMC counters cannot be read from inside this process, so the measured
            # traffic-per-algorithmic-byte ratio of this kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
            # passes, gfx950 correction applied — profiles/r04_pmc_traffic.md via tools/summarize_pmc.py, re-measured after the round-4 NTT change: 3.02) scales the launch's algorithmic bytes.
            traffic, traffic_source = None, None
            try:
                pmc = json.loads((ROOT / "profiles" / "pmc_traffic_lde.json").read_text())
                traffic = per_launch_bytes * float(pmc["traffic_bytes_per_alg_byte"])
                traffic_source = pmc["source"] + ": ratio x algorithmic bytes"
            except Exception:
                pass
            roof = {
                "kernel": "ntt2_pass_kernel, coset-LDE launches (each = 8 coset NTTs per column, 2 passes; wires 135 / Z+pp 20 / quotient 16 columns)",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "alg_bytes_per_launch": per_launch_bytes, "ms_per_launch": round(per_launch_ms, 4),
                "launches": lde["calls"],
            }
        # The dominant kernel by TIME is the Poseidon leaf hashing, which is integer-ALU bound (no HBM/MFMA roofline
        # applies): report its rate next to the instruction count measured with rocprofv3 --pmc SQ_INSTS_VALU.
        alu = None
        hl = prof.get("hash_leaves")
        if hl and hl["calls"] and args.workload == "prove":
            N = n << 3
            perms = N * (17 + 3 + 2)   # ceil(135/8) + ceil(20/8) + ceil(16/8) sponge permutations per LDE row
            ms = hl["ms"] / args.steps
            alu = {"kernel": "hash_leaves_colmajor_kernel", "bound": "integer ALU (VALU issue)", "perms_per_proof": perms,
                   "ms_per_proof": round(ms, 2), "gperms_per_s": round(perms / (ms * 1e-3) / 1e9, 3)}
            try:
                # per-permutation VALU instruction count: PMC-measured (SQ_INSTS_VALU), reproduced by the loop-weighted static
                # histogram of the gfx950 assembly (tools/alu_ceiling.py -> profiles/r04_alu_ceiling.json: re-derived after the round-4 hashing schedule)
                ceil_info = json.loads((ROOT / "profiles" / "r04_alu_ceiling.json").read_text())
                ipp = float(ceil_info["valu_insts_per_perm_pmc"] or ceil_info["valu_insts_per_perm_static"])
                achieved = perms / (ms * 1e-3) * ipp / 64.0
                # THE CEILING IS A BOUND: the fastest rate ANY measured instruction stream containing this kernel's
                # instruction classes reached (profiles/r03_ubench_int.md: `best_mixed_stream_cycles` per wavefront-instruction per
                # SIMD) at the device's MAXIMUM engine clock — one clock source, an upper limit of whatever the kernel ran at —
                # so frac <= 1 by construction.
                max_ghz = vx.lib().vx_device_max_clock_khz(local_rank) / 1e6
                if not 1.0 < max_ghz < 4.0:
                    raise RuntimeError(f"implausible device clock {max_ghz} GHz")
                c_best = float(ceil_info["best_mixed_stream_cycles"])
                ceiling = ceil_info["simds"] * max_ghz * 1e9 / c_best
                ghz = hash_clock_ghz           # sampled INSIDE the timed hash_leaves launches (s_memtime / s_memrealtime per sampled wave)
                quad = ceil_info["simds"] * ghz * 1e9 / 4.0
                alu.update({"valu_insts_per_perm": ipp, "achieved_wave_inst_per_s": achieved, "ceiling_wave_inst_per_s": ceiling,
                            "frac": round(achieved / ceiling, 4),
                            "ceiling_model": f"1024 SIMDs x device max clock {max_ghz:.2f} GHz / {c_best} cycles per wavefront-instruction (the best "
                                             "mixed slow-class stream tools/ubench_int.hip measured): a bound, not a fit",
                            # the round-2 figure, kept: one VALU instruction per wavefront per SIMD per QUAD-cycle at the clock sampled in
                            # the kernel (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU for this kernel); reads 1.03-1.09 because the in-kernel
                            # sample is ~4 % below the GRBM clock and a few fast-class instructions pair up
                            "frac_quad_issue_model": round(achieved / quad, 4), "clock_ghz_measured_in_kernel": round(ghz, 3),
                            "frac_at_pmc_run_clock": (round(achieved / (ceil_info["simds"] * float(ceil_info["pmc_run"]["clock_ghz"]) * 1e9 / 4.0), 4)
                                                      if (ceil_info.get("pmc_run") or {}).get("clock_ghz") else None),
                            "ubench_cycles_per_inst": ceil_info["cycles"],
                            "source": "profiles/r04_alu_ceiling.json (tools/alu_ceiling.py), profiles/r03_ubench_int.md, profiles/r04_pmc_sq_prove.md"})
            except Exception as e:  # the bench line must not die on a missing evidence file
                alu["ceiling_error"] = repr(e)
        sharded_mode = args.mode == "sharded" and world > 1 and args.workload == "prove"
        agg = H.aggregate(1 if sharded_mode else world, args.steps, dt)   # sharded: the N ranks finish ONE proof per step
        out = {
            "metric": metric, "value": agg["value"], "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": agg["ms_per_step"], "higher_is_better": True, "scaling": "strong" if sharded_mode else "weak",
            "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer modular arithmetic)", "data": "synthetic",
            "config": {"workload": wl_name,
                       "parallelism": (f"one proof coset-sharded x{world} (vx_prove_sharded; RCCL all-gathers of caps, quotient coset "
                                       f"coefficients, first FRI layer, openings; NO row-chunk all-to-all NTT in this mode — that "
                                       f"formulation exists for the commitment only: sharded.commit_sharded)" if sharded_mode else
                                       f"proof-level x{world} (one witness per GPU, no collective)")},
            "rccl_ranks": rccl_ranks, "rank_devices": rank_devices, "dist_backend": dist.get_backend() if dist is not None else None,
            "roofline": roof,
            "alu_bound_dominant_kernel": alu,
            "stage_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()},
            "stage_alg_GBps": {k: round(v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in prof.items()
                               if v["alg_bytes"] and v["ms"]},
        }
        if sharded_mode:
            ag = bench_prove._LEG["allgather"]
            out["exchange"] = {"allgather_calls_per_proof": ag.calls / args.steps, "inbound_bytes_per_rank_per_proof": ag.bytes / args.steps,
                               "backend": dist.get_backend(), "what": "in-place all-gathers vx_prove_sharded asked its host for (RCCL over xGMI)"}
        out["value_hbm_resident"] = out["value"]
        if host_leg is not None:
            out["value_end_to_end"] = host_leg["value"]          # witness in host memory -> proof bytes
            out["ms_per_step_end_to_end"] = host_leg["ms_per_step"]
            out["value_from_host_witness"] = host_leg
        if args.ranks_on_one_device:
            out["emulated_ranks_on_one_device"] = ("EVERY RANK IS A PROCESS ON DEVICE 0 over gloo: exercises the N > 1 code path on a single-GPU box; "
                                                   "no figure in this line is a multi-GPU measurement")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_commit(args) if args.workload == "commit" else bench_prove.cpu_baseline(args)

    # ---- THE CONTRACT'S LINE IS COMPLETE.  Everything below is optional legs, and ONE line goes to stdout whatever happens to them:
    #   * when they finish: the line with their results;
    #   * at the deadline (a worker that never comes up, a collective that never returns): a watchdog THREAD prints the line as it
    #     stands and ends the process — the main thread may be stuck inside a library call;
    #   * on SIGTERM (a launcher that lost a rank, a driver that ran out of patience): the C-level handler writes the signal number to a
    #     wake-up pipe at once, in whatever thread it lands; a second thread blocked on that pipe prints the line and ends the process —
    #     no Python-level handler has to wait for the main thread to come back from a collective.
    # The line lock is re-entrant and the line is built from a snapshot, so none of the three can deadlock or tear the others' output.
    import signal
    import threading
    line_lock = threading.RLock()
    state = {"printed": False}

    def emit_line(extra=None):
        """ONE line on stdout: the compact form (every number, no prose — the driver's record keeps the last 8 KB of stdout; the
        descriptions of the fields are in profiles/bench_line_glossary.md).  The complete line of the same run, prose included, goes to
        bench_line_full.json (in gpurun_out/ when that exists)."""
        with line_lock:
            if rank == 0 and not state["printed"]:
                d = dict(out)
                try:
                    d.update(dict(legs))      # whatever legs are complete by now (the watchdog / SIGTERM paths print before the merge below)
                except NameError:
                    pass
                if extra:
                    d.update(extra)
                try:
                    import bench_prove as bp
                    side = Path("gpurun_out") if Path("gpurun_out").is_dir() else Path(".")
                    (side / "bench_line_full.json").write_text(json.dumps(d))
                    line = json.dumps(d if args.full_line else bp.compact_line(d))
                except Exception as e:   # noqa: BLE001 — the line goes out whatever happens to its short form
                    d["compact_line_error"] = repr(e)
                    line = json.dumps(d)
                print(line, flush=True)
            state["printed"] = True

    def give_up(why):
        emit_line({"extra_legs": {"error": why + "; everything else in this line is complete"}})
        os._exit(0)

    multi = args.workload == "prove" and world > 1 and not args.no_multi_rank_legs and not args.circuit_flags
    deadline = args.multi_rank_leg_deadline if world > 1 else args.extra_legs_deadline
    watchdog = threading.Timer(deadline, lambda: give_up(f"the legs after the timed region did not finish within {deadline} s"))
    watchdog.daemon = True
    watchdog.start()
    wake_r, wake_w = os.pipe()
    os.set_blocking(wake_w, False)
    signal.signal(signal.SIGTERM, lambda *_: None)          # a handler must exist for the wake-up fd to be written; the work is done below
    signal.set_wakeup_fd(wake_w, warn_on_full_buffer=False)

    def on_signal():
        while True:
            b = os.read(wake_r, 1)
            if not b:
                return
            if b[0] == signal.SIGTERM:
                give_up("the process was told to terminate (SIGTERM) during the legs after the timed region")

    sig_thread = threading.Thread(target=on_signal, daemon=True)
    sig_thread.start()

    legs = {}
    t_legs = time.perf_counter()
    single = args.workload == "prove" and world == 1 and args.log_n >= 20 and not args.circuit_flags
    mr_spec = None
    if single and not args.no_dag_leg:
        from vectorx_amd import mapreduce as _mr
        mr_spec = _mr.DagSpec(*(int(x) for x in args.dag_spec.split(",")))

    def leg(name, fn):
        try:
            legs[name] = fn()
        except Exception as e:   # noqa: BLE001 — an extra leg must never cost the line
            legs[name] = {"error": repr(e)}

    if single and not args.no_dag_leg:
        if dag_pool is not None:
            ctx.trim()           # the workers need the HBM this process's buffer pool is sitting on
            leg("dag_header_range_512", lambda: bench_prove.dag_pool_legs(dag_pool, with_starks=not args.no_dag_stark_leg))
            both = legs["dag_header_range_512"]
            if "error" not in both:
                legs.update(both)
        else:
            leg("dag_header_range_512", lambda: bench_prove.dag_leg(ctx, local_rank))
            if not args.no_dag_stark_leg:
                leg("dag_header_range_512_with_starks", lambda: bench_prove.dag_with_starks_leg(ctx, local_rank, table_mode=args.dag_table_mode))
    if single and dag_pool is not None:
        dag_pool.close()         # the workers' HBM goes back before the legs this process proves itself
        dag_pool = None
    if single and not args.no_chip_leg:
        leg("chip_starks", lambda: bench_prove.chip_leg(ctx))
    if single and not args.no_rotate_leg:
        leg("rotate", lambda: bench_prove.rotate_leg(ctx, local_rank))
    if single and not args.no_dag_leg:
        # the DAG's three circuit sizes proven alone, recursion-shaped gate mix: ms per proof and the quotient by kernel / by gate
        leg("recursion_circuits_alone", lambda: bench_prove.recursion_profile_leg(ctx, mr_spec))
    if single and not args.no_cpu_baseline and not args.no_cpu_baseline_full and "cpu_baseline" in out:
        # LAST: the CPU baseline measured at the bench size, no scaling (~4 min of host cores; the GPU idles).  A child process: whatever
        # happens to it, the line goes out — with the bounded sample's figure if this one is missing.
        try:
            left = max(60.0, deadline - (time.perf_counter() - t_legs) - 30.0)
            full = bench_prove.cpu_baseline_full(args, timeout_s=left)
            cb = out["cpu_baseline"]
            sampled = {k: cb[k] for k in ("value", "seconds", "log_n", "stages_s") if k in cb}
            sampled["scaled_by"] = 1 << (args.log_n - cb.get("log_n", args.log_n))
            cb.update({"value": full["value"], "seconds": full["seconds"], "log_n": full["log_n"], "stages_s": full["stages_s"],
                       "cores": full["cores"], "proof_sha256": full["proof_sha256"], "sample": "measured in this run, no scaling",
                       "measured_in_this_run": True, "sampled": sampled})
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"]["full_size_error"] = repr(e)[:200]
            out["cpu_baseline"]["measured_in_this_run"] = False

    sharded_leg = dag_n_leg = dag_n_stark_leg = None
    if multi:
        dev = None if on_host else torch.device("cuda", local_rank)
        sharded_leg = bench_prove.guarded_collective_leg(dist, lambda: bench_prove.sharded_one_proof_leg(ctx, args, rank, world, dist, dev, sync))
        dag_n_leg = bench_prove.guarded_collective_leg(dist, lambda: bench_prove.dag_leg_ranks(ctx, args, local_rank, dist, dev))
        if not args.no_dag_stark_leg:
            dag_n_stark_leg = bench_prove.guarded_collective_leg(dist, lambda: bench_prove.dag_leg_ranks(ctx, args, local_rank, dist, dev, with_starks=True))
    pool_all = None
    if multi and pool_multi:
        # the same DAG on ONE pool of worker processes over all N GPUs: every rank hands its cached device buffers back and waits on the
        # host; rank 0 coordinates (jobs out / 32-byte digests back over a unix socket; no collective)
        cleanup()
        cleanup = lambda: None      # noqa: E731
        ctx.trim()
        sync()
        dist.barrier(group=host_group)
        if rank == 0:
            try:
                pool_all = bench_prove.dag_pool_legs(dag_pool, with_starks=not args.no_dag_stark_leg)
                pool_all["devices"] = list(pool_devices)
            except Exception as e:   # noqa: BLE001
                pool_all = {"error": repr(e)}
        dist.barrier(group=host_group)

    watchdog.cancel()
    if rank == 0:
        out.update(legs)
        if sharded_leg is not None:
            out["sharded_one_proof"] = sharded_leg
        if dag_n_leg is not None:
            out["dag_header_range_512"] = dag_n_leg
        if dag_n_stark_leg is not None:
            out["dag_header_range_512_with_starks"] = dag_n_stark_leg
        if pool_all is not None:
            out["dag_on_one_pool_over_all_gpus"] = pool_all
    emit_line()
    signal.set_wakeup_fd(-1)
    if dag_pool is not None:
        dag_pool.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    cleanup()      # free device objects BEFORE the context goes away
    ctx.close()


if __name__ == "__main__":
    main()

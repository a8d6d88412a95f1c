#!/usr/bin/env python3
"""bench.py -- T10 element-updates/s per Newton iteration on MI355X (BASELINE.json metric).

One "step" = ONE full Newton iteration of the implicit step on the workload, inputs resident in HBM:
  residual (F,P,f_int) -> gradient + norm -> tangent blocks -> row assembly of H -> PCG solve of H dv = -g
  -> v += dv, x = x_prev + h v         (SyncedNewton.cu:1046-1119); every 3rd iteration starts a new time step.
value = elements * steps / seconds (whole job, max over ranks).  The JSON line also carries
  roofline     : dominant kernel, algorithmic bytes per launch / mean launch duration (hipEvents on the launch stream)
  roofline_all : the same for every hot kernel
  cpu_baseline : the CPU oracle (C restatement of the reference, OpenMP) on the same workload, rank 0, N=1 only.

Launch:  python bench.py [--gpus N --steps K --warmup W --config C|B]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`, spawned before this process imports torch or touches the GPU) and relays rank 0's line.
Default workload = config C (972 000 T10 elements, the largest single-GPU T10 configuration of BASELINE.json); with N
GPUs every rank owns one config-C slab of a bar N times as long (N = 8: config E).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s, ~6.3 achievable)


def alg_bytes(E, N, nnz_coef, S=10, Q=5, cheb_bits=64, cheb_vec_bits=64):
    """Algorithmic (compulsory) bytes per launch of each hot kernel, see DESIGN.md section 4.
    S shape functions / Q quadrature points per element (T10: 10/5, ANCF-3243: 8/12, ANCF-3443: 16/48)."""
    npair = S * (S + 1) // 2
    elem_in = 4 * S + 24 * S * Q + 8 * Q  # connectivity + gradients + det J
    return {
        # inputs + x gather (unique coefficient vectors) -> 3S force components per element
        "residual": E * (elem_in + 24 * S) + N * 24,
        # + element-major symmetric block buffer (S(S+1)/2 blocks x 72 B)
        "tangent_blocks": E * elem_in + N * 24 + E * npair * 72,
        # block buffer in, scatter map in, mass in, H (9 doubles per coefficient pair) out once
        "assemble_rows": E * npair * 72 + E * 4 * S * S + nnz_coef * 8 + nnz_coef * 72,
        # fused tangent + assembly (T10, SVK): grad N + det J + per-point F in (each once: the re-reads by the owners of
        # an element's other rows are L2 traffic), instance map (code + S packed words), mass in, H out once
        "assemble_direct": E * (24 * S * Q + 8 * Q + 72 * Q) + E * S * (4 + 4 * S) + nnz_coef * 8 + nnz_coef * 72,
        # affine-element form (straight-sided T10): the element's 128-byte vertex-gradient record and 80 bytes of F per
        # point in (each once), instance map (8-byte header + four 8-byte words), H out once; grad N, det J and the
        # mass values are not read
        "assemble_affine": E * (128 + 80 * Q) + E * S * (8 + 32) + nnz_coef * 72,
        # H values + node-level columns + z,p_old in, p_new,q out
        "spmv": nnz_coef * 72 + nnz_coef * 4 + N * 96,
        # Chebyshev step: matrix (9 entries per block at cheb_bits: H itself or its scaled fp32/fp16 copy) + columns +
        # d_old gather, res in/out, D^-1, z in/out, d_new out
        "cheb_step": nnz_coef * 9 * cheb_bits // 8 + nnz_coef * 4 + N * (24 + 48 + 72 + 48 + 24) * cheb_vec_bits // 64,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", default="C",
                    help="C (default): T10 bar, 972 000 elements per GPU; B: T10 cube, 10 368 elements; D/A: ANCF")
    ap.add_argument("--rel-tol", type=float, default=1e-12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-pcg", type=int, default=50000, help="cap on PCG iterations (kernel experiments only)")
    ap.add_argument("--cheb-deg", type=int, default=0, help="Chebyshev preconditioner degree (1 = block-Jacobi, 0 = auto)")
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="device warm-up before the W warm-up steps: untimed Newton iterations for this many seconds "
                         "(clock ramp / first-touch events of a fresh process), then the state is reset")
    ap.add_argument("--local-precond", action="store_true",
                    help="multi-GPU: rank-local polynomial preconditioner (fewer collectives, 3-4x more CG iterations)")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="multi-GPU: exchange through the torch.distributed callback (~15-29 us of host time per collective) "
                         "instead of the engine's built-in RCCL all-reduce (enqueued from C++ on the solver's stream, ~1.6 us; "
                         "the default with the nccl backend; falls back to the callback if its communicator cannot be built)")
    ap.add_argument("--boundary-sums", action="store_true",
                    help="multi-GPU: the round-2 exchange (boundary rows summed over a global interface list, ~60 all-reduces "
                         "per CG iteration) instead of the overlapping partition (owner-computes with ghost layers, neighbour "
                         "exchanges: <= 12 per CG iteration)")
    ap.add_argument("--halo-depth", type=int, default=int(os.environ.get("TLFEA_HALO_DEPTH", "0")),
                    help="multi-GPU: ghost layers of the overlapping partition (one neighbour exchange buys depth - 1 coarse "
                         "steps; the overlap is computed redundantly).  0 = auto: 12 for x-slabs, 6 for 2-D / 3-D block grids")
    ap.add_argument("--slabs", action="store_true",
                    help="multi-GPU: x-slabs of a bar N times as long (the round-2 series: conditioning grows with N^2) instead of "
                         "the block series (2x1x1, 2x2x1, 2x2x2 blocks of the config: N = 8 is BASELINE's config E)")
    ap.add_argument("--precond", type=int, default=0, choices=(0, 1, 2),
                    help="0 auto, 1 Chebyshev polynomial, 2 two-level p-multigrid (T10, one GPU)")
    ap.add_argument("--linsolve-method", type=int, default=0, choices=(0, 1),
                    help="0 preconditioned CG (the benchmarked default), 1 the sparse direct solve (experiments; one GPU)")
    ap.add_argument("--cheb-kappa", type=float, default=0.0, help="polynomial interval [lmax/kappa, lmax] (0 = default)")
    ap.add_argument("--cheb-bits", type=int, default=0, choices=(0, 16, 32, 64),
                    help="matrix precision streamed by the Chebyshev steps (0 = auto = fp16 scaled copy)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Self-launch: one fresh child process per GPU through torch.distributed.run.  Nothing in THIS process has
        # imported torch or touched the GPU yet, and it never will: it only relays the children's output.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # TLFEA_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs,
    # collectives go through a host staging copy); production is nccl == RCCL over xGMI, one GPU per rank.
    backend = os.environ.get("TLFEA_BENCH_BACKEND", "nccl")
    n_dev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % n_dev if backend != "nccl" else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    tl = importlib.import_module("total-lagrangian-fea_amd")
    from importlib import import_module
    wl = import_module("total-lagrangian-fea_amd.workloads")
    par = import_module("total-lagrangian-fea_amd.partition")

    cfg = wl.CONFIGS[args.config]
    is_ancf = "kind" in cfg
    if is_ancf and world > 1:
        raise SystemExit("ANCF configs are single-GPU workloads")
    nx = 0 if is_ancf else cfg["cells"][0]
    # weak scaling: every rank owns one full-size x-slab of a bar `world` times as long
    part = hp = None
    halo = world > 1 and not args.boundary_sums
    if halo:
        # overlapping partition: the rank's block plus `halo_depth` ghost layers towards each neighbour, built rank-locally
        pgrid = (world, 1, 1) if args.slabs else par.process_grid(world)
        if args.halo_depth <= 0:
            args.halo_depth = 12 if pgrid[1] * pgrid[2] == 1 else 6
        w, hp = par.halo_block_structured(wl, args.config, rank, world, args.halo_depth, grid=pgrid)
    else:
        w = wl.build(args.config) if is_ancf else wl.build(args.config, cells=cfg["cells"], x_offset_cells=rank * nx)
    if world > 1 and not halo:
        par.restrict_bcs_to_global_ends(w, rank, world, cfg)
        lx = cfg["size"][0]
        part = par.slab_partition_structured(w["X"], lx * rank, lx * (rank + 1), rank, world)
        w["f_ext"] = (w["f_ext"].reshape(-1, 3) * part.node_weight[:, None]).reshape(-1)  # this rank's share
    d, s = wl.make_engine(tl, w)
    # a capped iteration count (--max-pcg, kernel experiments) accepts iterates above rel_tol; the default fails loudly
    s.SetLinSolveOpts(tl.LinSolveOpts(args.rel_tol, args.max_pcg, 25, args.cheb_deg, args.cheb_kappa, args.cheb_bits,
                                      args.precond, args.linsolve_method, int(args.max_pcg != 50000)))
    if world > 1:
        comm = None
        if backend == "nccl" and not args.torch_collectives and not os.environ.get("TLFEA_BENCH_TORCH_COLLECTIVES"):
            # every rank must take the same path: agree on whether ALL ranks can resolve RCCL inside the engine BEFORE any
            # rank enters ncclCommInitRank (decided here, never after: a rank waiting in the init for a peer that went
            # the other way would hang).  From then on a watchdog guards the init and a known-answer all-reduce + ring
            # send/recv: a rank that does not get the right answers within the limit prints why and exits non-zero.
            try:
                import ctypes
                binding = import_module("total-lagrangian-fea_amd.binding")
                probe_ok = binding.load_library().tlfea_rccl_unique_id(ctypes.create_string_buffer(128)) == 0
            except Exception:   # noqa: BLE001
                probe_ok = False
            okf = torch.tensor([1 if probe_ok else 0], device="cuda")
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 1:
                comm = par.rccl_communicator(dist, rank, world, timeout_s=float(os.environ.get("TLFEA_RCCL_TIMEOUT", "120")))
            elif rank == 0:
                print("built-in RCCL exchange unavailable on some rank, using the torch.distributed callback", file=sys.stderr)
        exchange_path = "built-in RCCL (C++)" if comm is not None else f"torch.distributed callback ({backend})"
        if halo:
            par.attach_halo(s, hp, torch, dist, native_rccl=comm)
        else:
            par.attach(s, part, torch, dist, local_preconditioner=args.local_precond, native_rccl=comm)
    d.UpdatePositions(w["x0"][:, 0], w["x0"][:, 1], w["x0"][:, 2])
    if os.environ.get("TLFEA_BENCH_SOLVER_VERBOSE"):
        s.SetVerbose(1)
    E, N = w["conn"].shape[0], w["X"].shape[0]
    E_local = E
    if halo:   # throughput counts the slab's own elements; the overlap is redundant work
        E = int(np.prod(cfg["cells"])) * 6
    nnz_coef = int(d.RetrieveMassCSRToCPU()[0][-1])

    step_ms = []

    def run(k, count_from=0):
        its = []
        for i in range(k):
            t_it = time.perf_counter()
            if (count_from + i) % 3 == 0:
                s.BeginStep()
            ng, it = s.NewtonIteration()   # returns ||g||: the host has the result, i.e. the iteration is complete
            its.append(it)
            step_ms.append(round((time.perf_counter() - t_it) * 1e3, 3))
        return its

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # A fresh process shows a few 30-80 ms stalls in its first ~0.3 s of GPU work (clock ramp, first-touch page
    # faults of the big buffers): keep the device busy for --prewarm-s, then start over from the initial state.
    n_prewarm = 0
    if args.prewarm_s > 0:
        t_pw = time.perf_counter()
        # multi-rank: a fixed count, every rank must run the same number of iterations (collectives inside)
        while (n_prewarm < 12) if world > 1 else (time.perf_counter() - t_pw < args.prewarm_s or n_prewarm < 3):
            run(1, n_prewarm)
            n_prewarm += 1
        d.UpdatePositions(w["x0"][:, 0], w["x0"][:, 1], w["x0"][:, 2])
        s.Setup()
        step_ms.clear()
    run(args.warmup)
    barrier()
    c0 = s.Collectives()
    cs0 = s.GetCommStats()
    t0 = time.perf_counter()
    pcg_its = run(args.steps, args.warmup)
    barrier()
    dt = time.perf_counter() - t0
    n_coll = s.Collectives() - c0
    cs1 = s.GetCommStats()
    lin_status = s.GetLinSolveStatus()
    if world > 1:
        t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = E * world * args.steps / dt
    if os.environ.get("TLFEA_BENCH_VERBOSE") and rank == 0:
        print("per-iteration ms (warm-up first):", step_ms, "CG iterations:", pcg_its, file=sys.stderr)

    # ---- per-kernel durations, live, hipEvents on the launch stream (separate pass: adds host syncs) ------
    s.SetProfiling(True)
    s.GetStageMs(reset=True)
    nprof = max(3, min(args.steps, 9))
    cs_prof0 = s.GetCommStats()
    run(nprof, args.warmup + args.steps)
    torch.cuda.synchronize()
    cs_prof1 = s.GetCommStats()
    st = s.GetStageMs(reset=True)
    s.SetProfiling(False)
    deg_eff, bits_eff, vec_bits = s.GetLinSolveInfo()
    ab = alg_bytes(E_local, N, nnz_coef, d.S, d.Q, bits_eff, vec_bits)
    fused = s.GetAssemblyMode() >= 2  # one fused tangent + assembly launch instead of tangent_blocks + assemble_rows
    fkey = "assemble_affine" if s.GetAssemblyMode() == 3 else "assemble_direct"
    if fused:   # the residual launch also writes what the fused assembly stages: F at every point
        ab["residual"] += E_local * (80 if fkey == "assemble_affine" else 72) * d.Q   # (general form: 72 B, affine form: 80 B per point)
    # mean launch duration: `reps` back-to-back launches per kernel between one hipEvent pair on the launch stream
    # (kernel time + same-stream boundary; agrees with rocprofv3 --kernel-trace, profiles/*kernel_stats.csv)
    kt = s.TimeKernels(reps=40 if E < 200000 else 10)
    if fused:
        kt[fkey] = kt.pop("assemble_rows")
        st = dict(st, **{fkey: st["assemble_rows"]})
        st.pop("assemble_rows")
        st["tangent_blocks"] = (0.0, 0)
    roof_all = {}
    n_outer = st["spmv"][1] // max(1, deg_eff)       # stage counter tallies deg launches per outer iteration
    pmg = s.GetPmgInfo()                              # (coarse nodes, coarse blocks, coarse polynomial degree) | None
    if pmg:   # V-cycle: 4 fine-level steps (2-term smoother before and after) + kc-1 coarse-level steps per CG iteration
        ab["cheb_step_coarse"] = pmg[1] * (9 * bits_eff // 8 + 4) + pmg[0] * (24 + 48 + 72 + 48 + 24) * vec_bits // 64
        # The polynomial-step kernel runs as three instantiations per CG iteration: fine level, non-final (3 launches: the
        # `cheb_step` entry, the iteration's largest share), fine level, final (1 launch, writes z in fp64 and the r.z
        # slots; same matrix stream, timed with the non-final ones), coarse level (kc - 1 launches).  They are reported
        # separately: the coarse level's 74 MB per launch live in the 256 MB Infinity Cache, so ITS rate is cache
        # bandwidth and must not be averaged into an HBM fraction.
        cyc = s.GetPmgCycleInfo()
        # two levels: kc - 1 vertex-level steps; three levels: 2 x vertex_terms smoothing passes on the vertex level and a
        # degree-k3 polynomial on the (few thousand) rigid-body-mode nodes below it
        ncs = 2 * cyc["vertex_terms"] if cyc["levels"] == 3 else pmg[2] - 1
        st = dict(st, spmv=(st["spmv"][0], n_outer), cheb_step=(0.0, n_outer * (2 * cyc["fine_terms"] - 1)),
                  cheb_step_coarse=(0.0, n_outer * ncs))
    else:
        st = dict(st, spmv=(st["spmv"][0], n_outer), cheb_step=(0.0, n_outer * (deg_eff - 1)))
    for k in ("residual", "tangent_blocks", "assemble_rows", "assemble_direct", "assemble_affine", "spmv", "cheb_step",
              "cheb_step_coarse"):
        if k not in st:
            continue
        ms, n = st[k]
        if n == 0:
            continue
        avg_s = kt[k] * 1e-3
        ach = ab[k] / avg_s / 1e9
        roof_all[k] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_us": round(avg_s * 1e6, 2),
                       "launches": n, "alg_bytes": ab[k], "total_ms": round(kt[k] * n, 3)}
    pmc, pmc_file = load_pmc_traffic(args.config) if world == 1 else ({}, None)
    for k, v in roof_all.items():
        if k in pmc:
            v["traffic"] = pmc[k]
            v["traffic_note"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, {pmc_file}): (2*FETCH_SIZE + "
                                 "WRITE_SIZE)*1024 B per launch; FETCH_SIZE counts 64 B per 128-B request on gfx950, calibrated "
                                 "for this engine's 8-/16-byte and scattered-row loads with tools/microbench/pmc_calib.hip")
    if pmg and "cheb_step" in roof_all and "cheb_step_coarse" in roof_all:   # (absent with --linsolve-method 1)
        roof_all["cheb_step"]["note"] = ("fine level of the V-cycle, non-final instantiation: 3 launches per CG iteration "
                                         "(the 4th fine pass is the final instantiation, same stream, not in `launches`)")
        roof_all["cheb_step_coarse"]["note"] = ("coarse level of the V-cycle: 74 MB per launch, resident in the 256 MB Infinity "
                                                "Cache -- cache bandwidth, NOT an HBM fraction; listed for completeness")
        roof_all["cheb_step_coarse"]["bound"] = "infinity-cache"
    # dominant = the single kernel instantiation with the largest total time (the cache-resident coarse step is not an
    # HBM-roofline candidate)
    dominant = max((k for k in roof_all if k != "cheb_step_coarse"), key=lambda k: roof_all[k]["total_ms"])
    roofline = dict(roof_all[dominant], kernel=dominant)
    if args.linsolve_method == 1:
        roofline["note"] = ("--linsolve-method 1: the step is the multifrontal factorisation (fp64 FMA bound, see `linear_solver`); the "
                            "kernel priced here is the largest of the HBM-bound launches that remain")
    elem_keys = ("residual", "grad", fkey) if fused else ("residual", "grad", "tangent_blocks", "assemble_rows")
    elem_ms = sum(st[k][0] for k in elem_keys) / nprof
    b_alg = 4 * d.S + 24 * d.S * d.Q + 8 * d.Q + 24 * d.S + 72.0 * nnz_coef / E + 24.0 * N / E
    stage_share = {k: round(st[k][0] / nprof, 4) for k in elem_keys + ("pcg", "update")}

    comm_report = {}
    if world > 1:
        n_cg = max(1, cs1["cg_iterations"] - cs0["cg_iterations"]) if halo else max(1, int(np.sum(pcg_its)))
        dx, da = cs1["exchanges"] - cs0["exchanges"], cs1["allreduces"] - cs0["allreduces"]
        comm_report = {
            "partition": (f"overlapping: owner-computes with {args.halo_depth} ghost layers per cut ({E_local - E} redundant "
                          f"elements on this rank), neighbour refreshes + 2 fixed-size all-reduces per CG iteration" if halo else
                          "boundary sums over the global interface list (one all-reduce per polynomial step)"),
            "neighbour_exchanges_per_cg_iteration": round((cs1["exchanges_in_cg"] - cs0["exchanges_in_cg"]) / n_cg, 2),
            "allreduces_per_cg_iteration": round((cs1["allreduces_in_cg"] - cs0["allreduces_in_cg"]) / n_cg, 2) if halo
            else round(n_coll / n_cg, 2),
            "exchanges_per_newton_iteration_outside_cg": round((dx + da - (cs1["exchanges_in_cg"] - cs0["exchanges_in_cg"]) -
                                                                (cs1["allreduces_in_cg"] - cs0["allreduces_in_cg"])) / args.steps, 2),
            "bytes_per_neighbour_exchange": round((cs1["bytes_exchanged"] - cs0["bytes_exchanged"]) / max(1, dx), 1),
            "bytes_per_allreduce": round((cs1["bytes_allreduced"] - cs0["bytes_allreduced"]) / max(1, da), 1),
            # profiling pass (eager launches, hipEvent pair around every exchange / all-reduce on the launch stream, rank 0)
            "comm_ms_per_step": round((cs_prof1["comm_ms"] - cs_prof0["comm_ms"]) / nprof, 4),
            "comm_note": "comm_ms_per_step: device time inside neighbour exchanges and all-reduces per Newton iteration on "
                         "rank 0, eager profiling pass; the timed region replays them inside hipGraphs with the built-in "
                         "RCCL exchange",
        }

    out = {
        "metric": "T10-tet element-updates/sec per Newton step", "value": round(value, 1), "unit": "element-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"config {args.config}: {w['desc']}, {E} elements / {N} nodes per GPU, " +
                               (f"implicit Newton iteration incl. PCG solve (rel_tol {args.rel_tol:g})" if args.linsolve_method == 0
                                else "implicit Newton iteration incl. sparse direct solve") +
                               (f"; {world} blocks in a {'x'.join(str(v) for v in w['grid'])} grid = one body of "
                                f"{'x'.join(str(c * g) for c, g in zip(cfg['cells'], w['grid']))} cells ({E * world} elements"
                                f"{', BASELINE config E' if args.config == 'C' and world == 8 and not args.slabs else ''})"
                                if halo else ""),
                   "elements_per_gpu": E, "nodes_per_gpu": N, "hessian_nnz": 9 * nnz_coef,
                   "pcg_outer_iters_per_step": round(float(np.mean(pcg_its)), 1),
                   **({"exchange": exchange_path, "collectives_per_cg_iteration":
                       round(n_coll / max(1.0, float(np.sum(pcg_its))), 1)} if world > 1 else {}),
                   **(comm_report if world > 1 else {}),
                   "last_solve_rel_residual": float(lin_status["rel_res"]), "last_solve_converged": lin_status["converged"],
                   **({"linear_solver": "sparse direct (the engine's multifrontal Cholesky, --linsolve-method 1): the "
                                        "preconditioner entry below does not apply"} if args.linsolve_method == 1 else {}),
                   "preconditioner": ("block-Jacobi (3x3)" if deg_eff <= 1 else
                                      ((f"three-level p-multigrid V-cycle (T10 -> vertex mesh of {pmg[0]} nodes -> rigid-body modes "
                                        f"of aggregates, {cyc['level3_nodes']} nodes; Galerkin operators; Chebyshev smoothers of "
                                        f"{cyc['fine_terms']} / {cyc['vertex_terms']} terms on the fine / vertex level, degree-"
                                        f"{cyc['level3_degree']} polynomial on the third), " if cyc["levels"] == 3 else
                                        f"two-level p-multigrid V-cycle (T10 -> vertex mesh, {pmg[0]} coarse nodes, Galerkin "
                                        f"coarse operator, {cyc['fine_terms']}-term Chebyshev smoothers, degree-{pmg[2]} coarse "
                                        f"polynomial), ")
                                       if pmg else (f"Chebyshev degree {deg_eff} of the operator scaled by its 12 x 12 node "
                                                    f"blocks (L^-1 H L^-T, the four coefficient vectors of an ANCF node), "
                                                    if s.GetPolynomialInfo()["block"] == 12 else
                                                    f"Chebyshev degree {deg_eff} of block-Jacobi, ")) + "steps stream a "
                                      f"{'scaled fp%d copy of H' % bits_eff if bits_eff != 64 else 'fp64 H'} "
                                      f"with fp{vec_bits} work vectors; "
                                      "outer CG, residual and convergence test in fp64 on H")},
        "element_stage": {"ms_per_step": round(elem_ms, 4), "value": round(E / (elem_ms * 1e-3), 1),
                          # connectivity + grad N + det J + x gather + H written once + f_int
                          "alg_bytes_per_element": round(b_alg, 1),
                          "roofline_frac": round(E * b_alg / (elem_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                          "kernels": list(elem_keys),
                          "note": "residual+gradient+tangent+assembly only (no linear solve), profiling pass; "
                                  "roofline_frac = E * alg_bytes_per_element / time / 8 TB/s"},
        "stage_ms_per_step": stage_share,
        "prewarm": {"seconds": args.prewarm_s, "newton_iterations": n_prewarm,
                    "note": "untimed device warm-up before the W warm-up steps; state reset afterwards"},
        "roofline": roofline, "roofline_all": roof_all,
    }

    if is_ancf:
        out["metric"] = "ANCF element-updates/sec per Newton step (not the BASELINE metric: T10 is)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not is_ancf:
        out["cpu_baseline"] = cpu_baseline(w, args)
    del s
    d.Destroy()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def load_pmc_traffic(config):
    """HBM-side bytes per launch from the committed PMC summary of THIS config (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes of the same command, tools/collect_profiles_r02.sh).  FETCH_SIZE is doubled:
    tools/microbench/pmc_calib.hip measures exactly half the bytes for this engine's access shapes too (8- and 16-byte
    streaming loads, scattered 1200-byte rows read in 80-byte segments; profiles/r02_pmc_calibration.txt);
    WRITE_SIZE is exact."""
    import csv
    import re
    path = None
    for cand in (f"r03_config{config}_pmc_hbm.csv", f"r02_config{config}_pmc_hbm.csv", f"r01_config{config}_pmc_hbm.csv"):
        if os.path.exists(os.path.join(ROOT, "profiles", cand)):
            path = os.path.join(ROOT, "profiles", cand)
            break
    if path is None:
        return {}, None
    names = {r"residual_kernel": "residual", r"tangent_blocks_kernel": "tangent_blocks",
             r"assemble_rows_kernel": "assemble_rows", r"assemble_direct_kernel": "assemble_direct",
             r"assemble_affine_kernel": "assemble_affine",
             r"spmv_dir_dot_kernel": "spmv",
             r"cheb_step_kernel<false>": "cheb_step",                 # fp64 polynomial steps (cheb_bits 64)
             # fp32 recurrence, non-final steps (rocprofv3 leaves names with _Float16 arguments mangled)
             r"cheb32_kernel<[^,]+, \d+, false": "cheb_step", r"cheb32_kernelI\w+?_Li\d+ELb0E": "cheb_step"}
    out, acc = {}, {}
    poly = {}   # counter -> [sum KB, launches] over every non-final launch of the polynomial-step kernel (both levels)
    for r in csv.DictReader(open(path)):
        if re.search(r"cheb32_kernelI\w+?_Li\d+ELb0E|cheb32_kernel<[^,]+, \d+, false", r["kernel"]):
            pp = poly.setdefault(r["counter"], [0.0, 0])
            pp[0] += float(r["mean_KB"]) * int(r["launches"])
            pp[1] += int(r["launches"])
        for frag, key in names.items():
            if re.search(frag, r["kernel"]):
                # several rows can match (the polynomial step runs on both levels of the p-multigrid cycle, one row per
                # grid size; the SpMV has two instantiations): keep the one with the larger traffic
                c = acc.setdefault(key, {})
                c[r["counter"]] = max(c.get(r["counter"], 0.0), float(r["mean_KB"]))
    for key, c in acc.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out[key] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    return out, os.path.relpath(path, ROOT)


def cpu_baseline(w, args):
    """The CPU oracle (C restatement of the reference, OpenMP atomics for the scatter, Jacobi-PCG twin of the
    device solver) on a BOUNDED sample of the workload: Newton iterations on the host cores for ~10-30 s.
    Large workloads (config C: one iteration of all 972 000 elements takes ~1 min on 16 cores) are sampled by a
    thinner x-slab of the same bar -- same cell size, cross-section, material, clamp and load -- of <= ~30 000 elements."""
    from importlib import import_module

    from oracle import orc

    wl = import_module("total-lagrangian-fea_amd.workloads")
    sample = "the whole workload"
    cfg = wl.CONFIGS[args.config]
    if w["conn"].shape[0] > 40000 and "cells" in cfg:
        nx, ny, nz = cfg["cells"]
        nxs = max(2, min(nx, 30000 // (6 * ny * nz)))
        w = wl.build(args.config, cells=(nxs, ny, nz))
        sample = (f"x-slab {nxs}x{ny}x{nz} cells of the {nx}x{ny}x{nz} bar (same cell size, cross-section, material, "
                  f"clamp at x=0, load on the slab's end face)")
    m = w["material"]
    mat = (orc.svk(m["E"], m["nu"], rho0=m["rho0"]) if m["kind"] == "svk"
           else orc.mooney_rivlin(m["mu10"], m["mu01"], m["kappa"], rho0=m["rho0"]))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16, int(os.environ.get("OMP_NUM_THREADS", "16"))))  # the GPU box's CPU share
    X, conn = w["X"], w["conn"]
    o = orc.T10Oracle(X, conn, mat, fixed=w["fixed"], f_ext=w["f_ext"])
    o.calc_dndu_pre()
    o.calc_mass()
    o.x, o.y, o.z = (np.ascontiguousarray(w["x0"][:, i]) for i in range(3))
    prm = w["params"]
    h, rho = prm[6], prm[3]
    t0 = time.perf_counter()
    n_it = 0
    t_elem = 0.0
    while True:
        te = time.perf_counter()
        f_int = o.internal_force(o.v)
        g = o.grad_L(f_int, h, rho)
        ro, ci, val = o.assemble_hessian(h, rho, nthreads=cores)
        t_elem += time.perf_counter() - te
        dv, its = orc.solve_pcg(ro, ci, val, -g, rel_tol=args.rel_tol, max_iter=50000, nthreads=cores)
        o.v += dv
        # x = x_prev + h v  (x_prev = x0)
        o.x = w["x0"][:, 0] + h * o.v[0::3]
        o.y = w["x0"][:, 1] + h * o.v[1::3]
        o.z = w["x0"][:, 2] + h * o.v[2::3]
        n_it += 1
        el = time.perf_counter() - t0
        # about 10 s of CPU work (at least 3 iterations, at most 30 s): a bounded sample of the same workload
        if (el > 10.0 and n_it >= 3) or el > 30.0:
            break
        if n_it % 3 == 0:   # like the timed GPU loop: a new implicit step every third Newton iteration
            o.v[:] = 0.0
            o.x, o.y, o.z = (np.ascontiguousarray(w["x0"][:, i]) for i in range(3))
    return {"value": round(conn.shape[0] * n_it / el, 1), "unit": "element-updates/s", "cores": cores,
            "kind": "port", "element_stage_value": round(conn.shape[0] * n_it / t_elem, 1),
            "comparable_to_value": sample == "the whole workload",
            "comparability_note": ("the sample is a thinner slab of the same bar solved with Jacobi-PCG: its conditioning and "
                                   "iteration count differ from the benchmarked mesh, so value / cpu_baseline.value is NOT a "
                                   "like-for-like speed-up (one Newton iteration of the full 972 000-element mesh takes the "
                                   "oracle minutes on 16 cores); the per-element rate of the element stage is comparable")
            if sample != "the whole workload" else "same workload",
            "sample": f"{n_it} Newton iteration(s) of {sample} ({conn.shape[0]} elements, PCG rel_tol {args.rel_tol:g}, "
                      f"{its} iterations last solve) in {el:.1f} s"}


if __name__ == "__main__":
    main()

"""world_size>1 coverage of the partitioned path.  CPU: gloo + oracle engine.
GPU box: two ranks on the one GPU driving libtlfea_hip.so, gloo with a host staging copy."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import load_mesh, tl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
par = __import__("importlib").import_module("total-lagrangian-fea_amd.partition")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, extra, tmp_path, timeout=600):
    out = tmp_path / "report.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--out", str(out)] + extra
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize("world", [2, 3])
def test_partition_bookkeeping(world):
    X, conn = load_mesh("res2")
    owner = par.slab_owner(X, conn, world)
    assert sorted(np.bincount(owner).tolist())[0] >= conn.shape[0] // world - 1
    parts = [par.partition_from_global(X, conn, owner, r, world) for r in range(world)]
    assert sum(len(p.elem_ids) for p in parts) == conn.shape[0]
    wsum = np.zeros(X.shape[0])
    slot_of = {}
    for p in parts:
        assert np.array_equal(p.l2g[p.conn], conn[p.elem_ids])          # local connectivity maps back exactly
        np.add.at(wsum, p.l2g, p.node_weight)
        for n, s in zip(p.l2g[p.iface_nodes], p.iface_slots):
            assert slot_of.setdefault(int(n), int(s)) == int(s)           # same node -> same slot on all ranks
        assert np.all(p.node_weight[p.iface_nodes] <= 0.5) and p.n_global_iface == parts[0].n_global_iface
    assert np.allclose(wsum, 1.0)                                         # weights partition unity
    assert len(slot_of) == parts[0].n_global_iface
    f = np.random.default_rng(0).normal(size=3 * X.shape[0])
    tot = np.zeros_like(f).reshape(-1, 3)
    for p in parts:
        np.add.at(tot, p.l2g, p.share_of_nodal_vector(f).reshape(-1, 3))
    assert np.allclose(tot.reshape(-1), f)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_every_replicated_node_has_one_owner(world):
    X, conn = load_mesh("res2")
    owner = par.slab_owner(X, conn, world)
    owners = np.zeros(X.shape[0], dtype=np.int32)
    for r in range(world):
        p = par.partition_from_global(X, conn, owner, r, world)
        assert p.node_owned.shape == p.node_weight.shape and np.all(p.node_owned[p.node_weight == 1.0] == 1)
        owners[p.l2g[p.node_owned == 1]] += 1
    assert np.all(owners == 1)


def test_structured_slab_interfaces_agree():
    """bench.py's rank-local slab construction: neighbours must agree on slot order without communicating."""
    wl = __import__("importlib").import_module("total-lagrangian-fea_amd.workloads")
    world = 3
    cfg = wl.CONFIGS["S"]
    nx, lx = cfg["cells"][0], cfg["size"][0]
    coords = {}
    for r in range(world):
        w = wl.build("S", cells=cfg["cells"], x_offset_cells=r * nx)
        p = par.slab_partition_structured(w["X"], lx * r, lx * (r + 1), r, world)
        for n, s in zip(p.iface_nodes, p.iface_slots):
            key = tuple(np.round(w["X"][n], 9))
            assert coords.setdefault(int(s), key) == key
        assert np.all(p.node_weight[p.iface_nodes] == 0.5)
        lo = np.abs(w["X"][:, 0] - lx * r) < 1e-9
        assert np.all(p.node_owned[lo] == (0 if r > 0 else 1)) and np.all(p.node_owned[~lo] == 1)
    assert len(coords) == (world - 1) * (2 * cfg["cells"][1] + 1) * (2 * cfg["cells"][2] + 1)


def test_two_ranks_gloo_oracle_engine(tmp_path):
    rep = launch(2, ["--engine", "oracle", "--mesh", "box", "--steps", "2"], tmp_path)
    assert rep["ok"] and rep["n_iface"] > 0, rep


@pytest.mark.gpu
def test_two_ranks_hip_engine_one_gpu(tmp_path):
    rep = launch(2, ["--engine", "hip", "--mesh", "res2", "--steps", "1"], tmp_path)
    assert rep["ok"] and rep["n_iface"] > 0, rep
    # the partitioned path runs the two-level p-multigrid cycle (one exchange per polynomial step on either level,
    # one for the restriction, three for the CG's reductions and boundary rows): a few dozen collectives per CG iteration
    # at the coarse degree of a small mesh, and about as many CG iterations as the un-partitioned cycle needs
    assert rep["precond"] == 2, rep
    per_it = rep["collectives"] / max(1, rep["pcg_iters"])
    assert per_it < 4 + 12 + 1 + 3 + 6, rep          # fine steps + coarse degree (12 at this size) + restriction + CG + set-up share


@pytest.mark.gpu
def test_three_ranks_hip_engine_one_gpu(tmp_path):
    rep = launch(3, ["--engine", "hip", "--mesh", "box", "--steps", "1"], tmp_path)
    assert rep["ok"] and rep["precond"] == 2, rep


@pytest.mark.gpu
def test_partitioned_multigrid_keeps_the_iteration_count(tmp_path):
    """CG iterations of the partitioned p-multigrid cycle (2 ranks) against the same solves on one rank: the cycle is
    the same operator (boundary rows summed before every step), so the counts agree up to the convergence-test cadence."""
    two = launch(2, ["--engine", "hip", "--mesh", "res4", "--steps", "1"], tmp_path)
    one = launch(1, ["--engine", "hip", "--mesh", "res4", "--steps", "1"], tmp_path)
    assert two["ok"] and one["ok"] and two["precond"] == 2 and one["precond"] == 2, (one, two)
    assert two["newton"] == one["newton"]
    assert two["pcg_iters"] <= 1.2 * one["pcg_iters"] + 2 * two["newton"], (one["pcg_iters"], two["pcg_iters"])


@pytest.mark.gpu
def test_rank_local_preconditioner_needs_fewer_collectives(tmp_path):
    """Same parity with both multi-GPU preconditioner forms; the rank-local one (owners set; optional) exchanges
    once per CG iteration for the polynomial instead of once per polynomial step, at the price of more CG iterations."""
    loc = launch(2, ["--engine", "hip", "--mesh", "res2", "--steps", "1", "--precond", "local"], tmp_path)
    exc = launch(2, ["--engine", "hip", "--mesh", "res2", "--steps", "1", "--precond", "exchange"], tmp_path)
    assert loc["ok"] and exc["ok"], (loc, exc)
    # the exchanging form is the partitioned p-multigrid cycle (~20 collectives per CG iteration on this mesh, far fewer CG
    # iterations); the rank-local polynomial needs one per iteration plus the CG's own
    assert loc["collectives"] * 2 < exc["collectives"], (loc, exc)
    assert exc["pcg_iters"] < loc["pcg_iters"] < 6 * exc["pcg_iters"], (loc, exc)


@pytest.mark.gpu
def test_rccl_exchange_path_single_rank(tmp_path):
    """The production exchange (torch.distributed 'nccl' = RCCL, in place on the engine's device buffers, default-stream
    ordering, no host copy) cannot run with two ranks on the one GPU of the test box; with world_size 1 and a band of
    nodes declared an interface of multiplicity 1 every collective of the partitioned path still runs -- as an identity
    all-reduce -- and the step must reproduce the un-partitioned oracle."""
    rep = launch(1, ["--engine", "hip", "--mesh", "res2", "--steps", "2", "--backend", "nccl", "--fake-iface"], tmp_path)
    assert rep["ok"] and rep["n_iface"] == 40 and rep["collectives"] > 100, rep


@pytest.mark.gpu
def test_builtin_rccl_exchange_single_rank(tmp_path):
    """The opt-in exchange that needs no host language in the loop (tlfea_rccl_*: RCCL resolved at run time, collectives
    enqueued from C++ on the solver's stream): communicator creation from a shipped unique id, every exchange of the
    partitioned path as a 1-rank all-reduce, same step as the un-partitioned oracle."""
    rep = launch(1, ["--engine", "hip", "--mesh", "res2", "--steps", "2", "--backend", "nccl", "--fake-iface",
                     "--native-rccl"], tmp_path)
    assert rep["ok"] and rep["n_iface"] == 40, rep


@pytest.mark.gpu
def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver calls it: no torch.distributed.run around it, WORLD_SIZE unset -- the
    script starts its two ranks itself and relays rank 0's JSON line (gloo rehearsal backend on the one-GPU test box)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(TLFEA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--config", "S", "--max-pcg", "2000"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["collectives_per_cg_iteration"] > 0
    assert "p-multigrid" in out["config"]["preconditioner"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py --gpus 2 end to end (slab construction, interface attach, timing loop, JSON line) with the gloo
    rehearsal backend: both ranks share the one GPU of the test box; production uses nccl (RCCL)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
           "1", "--config", "S", "--max-pcg", "2000"]
    env = dict(os.environ, TLFEA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["elements_per_gpu"] == 6 * 4 * 3 * 2


# ---- overlapping partition (round 3): owner-computes with ghost layers, neighbour exchanges ----------------------------
def _incidence_counts(n_nodes, conn):
    c = np.zeros(n_nodes, dtype=np.int64)
    np.add.at(c, conn.reshape(-1), 1)
    return c


@pytest.mark.parametrize("world,partitioner,mesh,depth", [(2, "slab", "res2", 3), (3, "slab", "res4", 4),
                                                          (3, "rcb", "bunny", 3), (5, "rcb", "res4", 3)])
def test_halo_partition_bookkeeping(world, partitioner, mesh, depth):
    X, conn = load_mesh(mesh)
    eo = par.rcb_owner(X, conn, world) if partitioner == "rcb" else par.slab_owner(X, conn, world)
    assert np.bincount(eo, minlength=world).min() >= conn.shape[0] // world - 1
    no = par.node_owner_from_elements(X.shape[0], conn, eo, world)
    inc = _incidence_counts(X.shape[0], conn)
    hps = [par.halo_partition(X, conn, no, np.arange(X.shape[0]), r, world, depth) for r in range(world)]
    owners = np.zeros(X.shape[0], dtype=np.int32)
    for hp in hps:
        owners[hp.l2g[:hp.n_owned]] += 1
        assert np.all(np.diff(hp.layer) >= 0) and hp.layer[0] == 0 and hp.layer.max() <= depth   # owned first, then by layer
        assert np.array_equal(hp.l2g[hp.conn], conn[hp.elem_gids])                                # connectivity maps back
        assert np.array_equal(no[hp.l2g] == hp.rank, hp.layer == 0)
        # rows of every layer < depth are complete: all incident elements are local
        inc_loc = _incidence_counts(len(hp.l2g), hp.conn)
        inner = hp.layer < depth
        assert np.array_equal(inc_loc[inner], inc[hp.l2g[inner]])
        # layers are graph distances: a node of layer k > 0 shares an element with one of layer k - 1
        lay_min = hp.layer[hp.conn].min(axis=1)
        best = np.full(len(hp.l2g), 99)
        np.minimum.at(best, hp.conn.reshape(-1), np.repeat(lay_min, 10))
        assert np.all(best[hp.layer > 0] == hp.layer[hp.layer > 0] - 1)
        for k, p in enumerate(hp.peers):
            q = hps[p]
            kq = q.peers.index(hp.rank)
            assert np.array_equal(hp.l2g[hp.send[k]], q.l2g[q.recv[kq]])           # same nodes, same order, both sides
            assert np.array_equal(hp.send_layer[k], q.layer[q.recv[kq]])
            assert np.all(np.diff(hp.send_layer[k]) >= 0) and np.all(hp.layer[hp.send[k]] == 0)
        # every ghost is received from exactly one peer
        got = np.concatenate(hp.recv) if hp.recv else np.zeros(0, dtype=np.int32)
        assert np.array_equal(np.sort(got), np.arange(hp.n_owned, len(hp.l2g)))
    assert np.all(owners == 1)


def test_halo_structured_slabs_agree():
    """bench.py's rank-local construction of overlapped slabs: neighbours agree on every exchange list (same global
    lattice ids in the same order) without communicating; clamp and load live where the long bar has them."""
    wl = __import__("importlib").import_module("total-lagrangian-fea_amd.workloads")
    world, depth = 3, 3
    built = [par.halo_slab_structured(wl, "S", r, world, depth) for r in range(world)]
    cfg = wl.CONFIGS["S"]
    n_plane = (2 * cfg["cells"][1] + 1) * (2 * cfg["cells"][2] + 1)
    assert all(w["grid"] == (world, 1, 1) for w, _ in built)
    for r, (w, hp) in enumerate(built):
        assert hp.peers == [p for p in (r - 1, r + 1) if 0 <= p < world]
        assert w["X"].shape[0] == len(hp.layer) and w["conn"].max() < len(hp.layer)
        assert len(w["fixed"]) == (n_plane if r == 0 else 0)
        assert np.count_nonzero(w["f_ext"]) == (n_plane if r == world - 1 else 0)
        for k, p in enumerate(hp.peers):
            q = built[p][1]
            kq = q.peers.index(r)
            assert np.array_equal(hp.l2g[hp.send[k]], q.l2g[q.recv[kq]])
            assert np.array_equal(hp.send_layer[k], q.layer[q.recv[kq]])
            assert np.allclose(w["X"][hp.send[k]], built[p][0]["X"][q.recv[kq]])
            # a ghost copy starts where its owner starts (the timing state's noise is keyed by the node, not by a local index)
            assert np.array_equal(w["x0"][hp.send[k]], built[p][0]["x0"][q.recv[kq]])
    total_owned = sum(hp.n_owned for _, hp in built)
    assert total_owned == (2 * cfg["cells"][0] * world + 1) * n_plane


@pytest.mark.gpu
@pytest.mark.parametrize("world,mesh,partitioner,depth", [(2, "res4", "slab", 4), (3, "res4", "rcb", 5), (3, "bunny", "rcb", 4)])
def test_halo_hip_engine_matches_unpartitioned_oracle(tmp_path, world, mesh, partitioner, depth):
    """The overlapping partition on 2 / 3 ranks (sharing the one GPU of the test box, gloo with host staging): positions of
    owned AND ghost nodes equal the un-partitioned oracle's (1e-10 of the displacement), same outer / Newton counts; the
    cycle is the single-GPU operator, so the CG iteration count stays within 1.2x; and the exchange budget holds:
    neighbour refreshes + all-reduces per CG iteration <= 12 (VERDICT r02 #1b) -- here 1 + ceil((kc + ks) / depth) + 2."""
    # the bunny's ||g|| ends at the round-off floor of its h rho c terms (1e-4, the tolerance): there the NUMBER of Newton
    # iterations is noise in any implementation, positions are not
    loose = ["--loose-counts"] if mesh == "bunny" else []
    rep = launch(world, ["--engine", "hip", "--mesh", mesh, "--steps", "1", "--mode", "halo", "--depth", str(depth),
                         "--partitioner", partitioner] + loose, tmp_path)
    assert rep["ok"] and rep["precond"] == 2 and rep["n_iface"] > 0, rep
    one = launch(1, ["--engine", "hip", "--mesh", mesh, "--steps", "1"] + loose, tmp_path)
    assert one["ok"] and (rep["newton"] == one["newton"] or loose), (one, rep)
    if loose:
        return
    assert rep["pcg_iters"] <= 1.2 * one["pcg_iters"] + 2 * rep["newton"], (one["pcg_iters"], rep["pcg_iters"])
    c = rep["comm"]
    per_it = (c["exchanges_in_cg"] + c["allreduces_in_cg"]) / max(1, c["cg_iterations"])
    assert c["cg_iterations"] >= rep["pcg_iters"] and per_it <= 12.0, rep


@pytest.mark.gpu
def test_halo_builtin_rccl_single_rank(tmp_path):
    """The production exchange of the overlapping partition (tlfea_rccl_*: communicator from a shipped unique id under the
    init watchdog, known-answer self-check, ncclAllReduce / ncclSend / ncclRecv enqueued from C++ on the solver's stream)
    with the one rank the test box allows: no neighbour exists, but every all-reduce of the path runs -- CAPTURED in the
    CG iteration's hipGraphs -- and the step must reproduce the un-partitioned oracle."""
    rep = launch(1, ["--engine", "hip", "--mesh", "res2", "--steps", "2", "--backend", "nccl", "--mode", "halo",
                     "--native-rccl"], tmp_path)
    assert rep["ok"] and rep["precond"] == 2, rep
    c = rep["comm"]
    assert c["exchanges"] == 0 and c["allreduces_in_cg"] >= 2 * c["cg_iterations"] > 0, rep


@pytest.mark.parametrize("world", [4, 8])
def test_halo_structured_blocks_agree(world):
    """The block series of bench.py (2x2x1, 2x2x2 blocks of the config; 8 blocks of config C = BASELINE's config E): every
    rank builds its block alone, every pair of neighbours (faces, edges, corners: up to 7 peers) agrees on the exchange
    lists and on the start positions of shared nodes, and the owned sets tile the body's lattice exactly once."""
    wl = __import__("importlib").import_module("total-lagrangian-fea_amd.workloads")
    built = [par.halo_block_structured(wl, "S", r, world, 2) for r in range(world)]
    cfg = wl.CONFIGS["S"]
    pg = par.process_grid(world)
    assert sum(hp.n_owned for _, hp in built) == int(np.prod([2 * c * g + 1 for c, g in zip(cfg["cells"], pg)]))
    gids = np.concatenate([hp.l2g[:hp.n_owned] for _, hp in built])
    assert len(np.unique(gids)) == len(gids)
    for r, (w, hp) in enumerate(built):
        assert len(hp.peers) == world - 1 and w["grid"] == pg            # 2 blocks per axis: every block touches every other
        for k, p in enumerate(hp.peers):
            q = built[p][1]
            kq = q.peers.index(r)
            assert np.array_equal(hp.l2g[hp.send[k]], q.l2g[q.recv[kq]])
            assert np.array_equal(hp.send_layer[k], q.layer[q.recv[kq]])
            assert np.array_equal(w["x0"][hp.send[k]], built[p][0]["x0"][q.recv[kq]])
    n_fixed = sum(len(w["fixed"][hp.layer[w["fixed"]] == 0]) for w, hp in built)
    assert n_fixed == (2 * cfg["cells"][1] * pg[1] + 1) * (2 * cfg["cells"][2] * pg[2] + 1)   # the x = 0 face, owned once


@pytest.mark.parametrize("world,partitioner", [(2, "slab"), (3, "rcb"), (8, "rcb")])
def test_halo_gloo_oracle_engine(tmp_path, world, partitioner):
    """CPU coverage of the overlapping partition (no GPU): every rank runs the oracle on its overlapped sub-mesh; the
    exchange lists of partition.halo_partition drive gloo send / recv pairs (ghost refresh of the CG direction and of the
    Newton update), dot products run over owned DOFs; owned AND ghost positions equal the un-partitioned oracle's.
    8 ranks: three levels of coordinate bisection, i.e. the 2 x 2 x 2 arrangement of the scaling run (a rank exchanges with
    face, edge and corner neighbours)."""
    rep = launch(world, ["--engine", "oracle", "--mesh", "box", "--steps", "2", "--mode", "halo", "--depth", "2",
                         "--partitioner", partitioner], tmp_path)
    assert rep["ok"] and rep["n_iface"] > 0 and rep["max_dup"] <= 1e-12, rep


@pytest.mark.gpu
def test_bench_four_ranks_block_grid_rehearsal(tmp_path):
    """bench.py --gpus 4: the block series (2 x 2 x 1 blocks of the config, three peers per rank: faces and the shared
    edge) with the gloo rehearsal backend, four ranks on the one GPU of the test box."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup",
           "1", "--config", "S", "--max-pcg", "2000", "--halo-depth", "4"]
    env = dict(os.environ, TLFEA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["value"] > 0 and out["config"]["last_solve_converged"]
    assert "2x2x1 grid" in out["config"]["workload"] and out["config"]["neighbour_exchanges_per_cg_iteration"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,partitioner,depth", [(2, "slab", 8), (3, "rcb", 6)])
def test_halo_three_level_cycle_replicated_coarsest_level(tmp_path, monkeypatch, world, partitioner, depth):
    """The three-level cycle on the overlapping partition (forced on a small mesh): the third level lives on a regular grid
    of bins every rank derives from coordinates alone, is REPLICATED (H3 and the level-3 residual are summed over the ranks'
    owned rows with one all-reduce each, its polynomial runs with no exchange), the vertex level smooths with 6 terms
    between ghost refreshes.  Same positions as the un-partitioned oracle, CG iterations within 1.2x of one rank running
    the same cycle, exchange budget <= 12 per CG iteration."""
    monkeypatch.setenv("TLFEA_PMG_LEVELS", "3")
    monkeypatch.setenv("TLFEA_PMG_AGG", "bins")
    rep = launch(world, ["--engine", "hip", "--mesh", "res4", "--steps", "1", "--mode", "halo", "--depth", str(depth),
                         "--partitioner", partitioner], tmp_path)
    assert rep["ok"] and rep["precond"] == 2 and rep["pmg_levels"] == 3, rep
    one = launch(1, ["--engine", "hip", "--mesh", "res4", "--steps", "1", "--mode", "halo", "--depth", str(depth)], tmp_path)
    assert one["ok"] and one["pmg_levels"] == 3 and rep["newton"] == one["newton"], (one, rep)
    assert rep["pcg_iters"] <= 1.2 * one["pcg_iters"] + 2 * rep["newton"], (one["pcg_iters"], rep["pcg_iters"])
    c = rep["comm"]
    per_it = (c["exchanges_in_cg"] + c["allreduces_in_cg"]) / max(1, c["cg_iterations"])
    assert per_it <= 12.0, rep

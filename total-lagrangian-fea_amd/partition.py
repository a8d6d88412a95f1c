"""Mesh partitioning across the GPUs of one node (one process per GPU).

Elements are owned by exactly one rank; nodes on partition boundaries are replicated.  Per Newton iteration
the path exchanges ONLY partition-boundary data: the boundary entries of the gradient, of the diagonal blocks
and of each SpMV result are summed over ranks with one packed all-reduce over a global interface list (plus the
fixed-size reduction slots of the dot products).  RCCL over xGMI in production (device buffers, no host copy),
gloo through a host staging copy in the CPU/one-GPU tests.  Payload is tiny (config E: 0.7 MB per slab cut), so
the exchange is latency-bound and is fused to two collectives per CG iteration."""
import numpy as np


class Partition:
    """One rank's sub-mesh: local numbering, interface list and weights."""

    def __init__(self, rank, world, X_loc, conn_loc, l2g, iface_nodes, iface_slots, n_global_iface, node_weight,
                 elem_ids=None, node_owned=None):
        self.rank, self.world = rank, world
        self.X, self.conn, self.l2g = X_loc, conn_loc, l2g
        self.iface_nodes = np.ascontiguousarray(iface_nodes, dtype=np.int32)
        self.iface_slots = np.ascontiguousarray(iface_slots, dtype=np.int32)
        self.n_global_iface = int(n_global_iface)
        self.node_weight = np.ascontiguousarray(node_weight, dtype=np.float64)
        self.elem_ids = elem_ids
        # exactly one rank owns each replicated node (the lowest rank holding it): the rank-local preconditioner
        # works on the owned nodes only
        self.node_owned = (np.ones(len(self.node_weight), dtype=np.int32) if node_owned is None
                           else np.ascontiguousarray(node_owned, dtype=np.int32))

    def localize_nodes(self, global_nodes):
        """Global node ids -> local ids (dropping nodes this rank does not hold)."""
        g = np.asarray(global_nodes, dtype=np.int64)
        pos = np.searchsorted(self.l2g, g)
        pos[pos >= len(self.l2g)] = 0
        ok = self.l2g[pos] == g
        return pos[ok].astype(np.int32)

    def share_of_nodal_vector(self, f_global_3n):
        """This rank's share of a global nodal load vector (replicated nodes get weight 1/multiplicity)."""
        f = np.asarray(f_global_3n).reshape(-1, 3)[self.l2g]
        return (f * self.node_weight[:, None]).reshape(-1)


def slab_owner(X, conn, world, axis=0):
    """Element owner by centroid position along `axis`, equal element counts per rank."""
    c = X[conn[:, :4]].mean(axis=1)[:, axis]
    order = np.argsort(c, kind="stable")
    owner = np.empty(conn.shape[0], dtype=np.int32)
    bounds = np.linspace(0, conn.shape[0], world + 1).astype(np.int64)
    for r in range(world):
        owner[order[bounds[r]:bounds[r + 1]]] = r
    return owner


def rcb_owner(X, conn, world):
    """Element owner by recursive coordinate bisection of the element centroids (unstructured meshes): the
    set is split at the weighted median along its longest extent, `world` = any count (parts get sizes in proportion to
    the ranks they will hold); deterministic (stable sorts), balanced to one element."""
    c = X[conn[:, :4]].mean(axis=1)
    owner = np.zeros(conn.shape[0], dtype=np.int32)

    def split(ids, r0, nr):
        if nr == 1:
            owner[ids] = r0
            return
        ext = c[ids].max(axis=0) - c[ids].min(axis=0)
        ax = int(np.argmax(ext))
        order = ids[np.argsort(c[ids, ax], kind="stable")]
        nl = nr // 2
        cut = (len(ids) * nl) // nr
        split(order[:cut], r0, nl)
        split(order[cut:], r0 + nl, nr - nl)

    split(np.arange(conn.shape[0]), 0, world)
    return owner


def partition_from_global(X, conn, owner, rank, world):
    """General partition of a global T10 mesh by an element->rank map (bit-exact integer bookkeeping)."""
    N = X.shape[0]
    mult = np.zeros(N, dtype=np.int32)
    first = np.full(N, world, dtype=np.int32)                 # lowest rank holding each node = its owner
    for r in range(world):
        held = np.unique(conn[owner == r])
        mult[held] += 1
        first[held] = np.minimum(first[held], r)
    iface_global = np.where(mult > 1)[0]                      # sorted global ids == slot order on every rank
    elem_ids = np.where(owner == rank)[0]
    l2g = np.unique(conn[elem_ids])                           # sorted -> local numbering keeps global order
    conn_loc = np.searchsorted(l2g, conn[elem_ids]).astype(np.int32)
    is_if = mult[l2g] > 1
    iface_nodes = np.where(is_if)[0]
    iface_slots = np.searchsorted(iface_global, l2g[iface_nodes])
    return Partition(rank, world, X[l2g].copy(), conn_loc, l2g, iface_nodes, iface_slots, len(iface_global),
                     1.0 / mult[l2g], elem_ids, first[l2g] == rank)


def slab_partition_structured(X_loc, x_lo, x_hi, rank, world, tol=1e-9):
    """Partition data of one x-slab of a structured bar built rank-locally (bench.py weak scaling): the
    interface with rank-1 is the plane x=x_lo, with rank+1 the plane x=x_hi; planes are ordered by (z,y) so both
    neighbours agree on the slot order without any communication."""
    N = X_loc.shape[0]
    w = np.ones(N)
    owned = np.ones(N, dtype=np.int32)
    nodes, slots = [], []
    plane_size = None
    for side, xv, k in (("lo", x_lo, rank - 1), ("hi", x_hi, rank)):
        if (side == "lo" and rank == 0) or (side == "hi" and rank == world - 1):
            continue
        ids = np.where(np.abs(X_loc[:, 0] - xv) < tol)[0]
        order = np.lexsort((X_loc[ids, 1], X_loc[ids, 2]))     # z major, then y
        ids = ids[order]
        plane_size = len(ids)
        nodes.append(ids)
        slots.append(k * plane_size + np.arange(plane_size))
        w[ids] = 0.5
        if side == "lo":
            owned[ids] = 0                                     # the plane shared with rank-1 belongs to rank-1
    if not nodes:
        return Partition(rank, world, X_loc, None, np.arange(N), [], [], 0, w, node_owned=owned)
    return Partition(rank, world, X_loc, None, np.arange(N), np.concatenate(nodes), np.concatenate(slots),
                     (world - 1) * plane_size, w, node_owned=owned)


class _DevicePtr:
    """Zero-copy view of a raw device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def make_allreduce(torch, dist, backend):
    """-> (callable(ptr, n), sync_before_callback).  'nccl': in place on the device buffer (RCCL);
    anything else: host staging copy (gloo)."""
    if backend == "nccl":
        views = {}  # the engine reuses one exchange buffer with a handful of lengths: wrap each (ptr, n) once

        def ar(ptr, n):
            t = views.get((ptr, n))
            if t is None:
                t = views[(ptr, n)] = torch.as_tensor(_DevicePtr(ptr, n), device="cuda")
            dist.all_reduce(t)
        return ar, 0  # torch's default stream is the null stream the engine launches on

    def ar_host(ptr, n):
        t = torch.as_tensor(_DevicePtr(ptr, n), device="cuda")
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        torch.cuda.synchronize()
    return ar_host, 1


def rccl_communicator(dist, rank, world, timeout_s=120.0, self_check=True):
    """A RCCL communicator owned by the engine (tlfea_rccl_*): rank 0 creates the unique id, torch.distributed ships its
    128 bytes once, every rank joins -- under a watchdog: a rank whose ncclCommInitRank or whose known-answer check
    (all-reduce + ring send/recv, tlfea_rccl_self_check) does not finish within timeout_s, or sees a wrong answer, prints
    why and exits with code 97 (a stuck collective cannot be abandoned), so a broken fabric shows as a failed launch, not
    a hang.  -> opaque handle for SyncedNewtonSolver.SetHalo / SetInterfaceRccl / rccl_destroy."""
    import ctypes as C

    from . import binding
    lib = binding.load_library()
    buf = C.create_string_buffer(128)
    if rank == 0:
        binding.check(lib.tlfea_rccl_unique_id(buf))
    box = [buf.raw]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    comm = C.c_void_p()
    binding.check(lib.tlfea_rccl_comm_create_timeout(box[0], int(rank), int(world), C.c_double(timeout_s), C.byref(comm)))
    if self_check:
        binding.check(lib.tlfea_rccl_self_check(comm, int(rank), int(world), C.c_double(timeout_s)))
    return comm


def rccl_destroy(comm):
    from . import binding
    binding.check(binding.load_library().tlfea_rccl_comm_destroy(comm))


def attach(solver, part, torch, dist, local_preconditioner=False, native_rccl=None):
    """Wire a SyncedNewtonSolver to its partition's interface exchange.  local_preconditioner: rank-local polynomial
    preconditioner on the owned nodes (one collective per CG iteration for its result) instead of one exchange per
    polynomial step.  Off by default: without overlap the decoupled blocks cost 3-4x more CG iterations (measured:
    two config-B slabs 35 -> 96 iterations), which outweighs the saved collectives unless the
    fabric latency is far above the kernel times."""
    if native_rccl is not None:   # opt-in: the engine's own RCCL communicator, collectives enqueued from C++
        solver.SetInterfaceRccl(part.iface_nodes, part.iface_slots, part.n_global_iface, part.node_weight, native_rccl)
    else:
        ar, sync = make_allreduce(torch, dist, dist.get_backend())
        solver.SetInterface(part.iface_nodes, part.iface_slots, part.n_global_iface, part.node_weight, ar, sync)
    if local_preconditioner:
        solver.SetInterfaceOwners(part.node_owned)


def restrict_bcs_to_global_ends(w, rank, world, cfg):
    """Bar-style configs (clamp at the global x=0 face, load on the global x=L face): a slab keeps the clamp only
    if it holds the global x=0 face and the end load only on the last slab; cube-style BCs (z faces) exist on
    every slab and are kept."""
    X = w["X"]
    lx = cfg["size"][0]
    if cfg["material"] == "svk":
        if rank > 0:
            w["fixed"] = np.zeros(0, dtype=np.int32)
        w["f_ext"][:] = 0.0
        if rank == world - 1:
            face = np.where(np.abs(X[:, 0] - lx * world) < 1e-9)[0]
            w["f_ext"][3 * face] = 5000.0 / max(1, len(face))
    return w


# ---- overlapping partition (round 3): owner-computes with ghost layers -------------------------------------------------
# Every node has exactly ONE owner.  A rank holds its owned nodes plus `depth` layers of ghost nodes (layer k = nodes at
# graph distance k from the owned set, two nodes being adjacent when they share an element) and every element that
# touches a node of layer < depth, so the rows of H, M and grad L of all nodes of layers < depth are COMPLETE on the rank:
# nothing is summed over ranks.  What the path exchanges instead is a refresh of ghost VALUES from their owners
# (neighbour to neighbour, payload independent of the number of ranks), and only every few steps: after a refresh the
# ghost layers are exact, each SpMV-like step invalidates the outermost valid layer, so a halo of depth G buys G - 1
# steps of the coarse polynomial per exchange (computed redundantly on the overlap).  Dot products weigh owned nodes 1,
# ghosts 0, and are summed with one small world all-reduce.
class HaloPartition:
    """One rank's overlapped sub-mesh.  Local numbering: owned nodes first (original relative order), then ghosts by
    increasing layer.  For peer p: `send[p]` = my owned nodes that p holds as ghosts, `recv[p]` = my ghosts owned by p;
    both sorted by (layer on the receiving side, global id), so 'refresh up to layer D' is a PREFIX of either list
    (`send_upto[p][D]`, `recv_upto[p][D]`) and both sides agree on the order without communicating."""

    def __init__(self, rank, world, depth, X, conn, l2g, layer, peers, send, send_layer, recv, elem_gids, src=None):
        self.rank, self.world, self.depth = rank, world, int(depth)
        self.X, self.conn = X, conn
        self.l2g = l2g
        self.src = l2g if src is None else src      # index of each local node in the mesh halo_partition was given
        self.layer = np.ascontiguousarray(layer, dtype=np.int32)
        self.n_owned = int(np.sum(self.layer == 0))
        self.peers = [int(p) for p in peers]
        self.send = [np.ascontiguousarray(a, dtype=np.int32) for a in send]
        self.send_layer = [np.ascontiguousarray(a, dtype=np.int32) for a in send_layer]
        self.recv = [np.ascontiguousarray(a, dtype=np.int32) for a in recv]
        self.elem_gids = elem_gids
        self.node_owned = (self.layer == 0).astype(np.int32)

    def n_upto(self, k):
        """number of local nodes of layers <= k (a prefix of the local numbering)"""
        return int(np.searchsorted(self.layer, k, side="right"))

    def localize_nodes(self, global_nodes):
        """Global node ids -> local ids (dropping nodes this rank does not hold)."""
        g = np.asarray(global_nodes, dtype=np.int64)
        order = np.argsort(self.l2g, kind="stable")
        sl = self.l2g[order]
        pos = np.searchsorted(sl, g)
        pos[pos >= len(sl)] = 0
        ok = sl[pos] == g
        return np.sort(order[pos[ok]]).astype(np.int32)

    def local_nodal_vector(self, f_global_3n):
        """A global nodal vector on this rank's nodes, UNSHARED: every rank evaluates complete rows."""
        return np.ascontiguousarray(np.asarray(f_global_3n).reshape(-1, 3)[self.l2g]).reshape(-1)


def _node_layers(n_nodes, conn, seed_mask, max_layer):
    """BFS over 'shares an element': layer[n] = graph distance from the seed set (max_layer + 1 = farther)."""
    layer = np.full(n_nodes, max_layer + 1, dtype=np.int32)
    layer[seed_mask] = 0
    reached = seed_mask.copy()
    for k in range(1, max_layer + 1):
        touch = reached[conn].any(axis=1)              # elements with a node of layer < k
        nodes = np.unique(conn[touch])
        new = nodes[~reached[nodes]]
        if len(new) == 0:
            break
        layer[new] = k
        reached[new] = True
    return layer


def halo_partition(X, conn, node_owner, gid, rank, world, depth):
    """Overlapped sub-mesh of `rank` out of a mesh that contains at least everything within `depth` + 1 layers of the
    rank's owned nodes (the whole global mesh in the tests; a generously extended slab in bench.py).
    node_owner[n] = owning rank of node n, gid[n] = a global id all ranks agree on (ordering of the exchange lists)."""
    conn = np.asarray(conn)
    node_owner = np.asarray(node_owner)
    gid = np.asarray(gid, dtype=np.int64)
    N = X.shape[0]
    G = int(depth)
    owned = node_owner == rank
    layer = _node_layers(N, conn, owned, G)
    # elements that touch a node of layer < G: rows of layers < G are complete
    keep_e = (layer[conn] < G).any(axis=1)
    elem_ids = np.where(keep_e)[0]
    keep_n = np.zeros(N, dtype=bool)
    keep_n[conn[elem_ids]] = True
    assert np.all(layer[keep_n] <= G)
    old = np.where(keep_n)[0]
    order = np.lexsort((old, layer[old]))              # by layer, original order inside a layer
    old = old[order]
    new_of = np.full(N, -1, dtype=np.int64)
    new_of[old] = np.arange(len(old))
    conn_loc = new_of[conn[elem_ids]].astype(np.int32)
    lay_loc = layer[old]
    own_loc = node_owner[old]
    gid_loc = gid[old]
    peers = sorted(int(p) for p in np.unique(own_loc) if p != rank)
    send, send_layer, recv = [], [], []
    for p in peers:
        # my ghosts owned by p, by (my layer, gid)
        r = np.where(own_loc == p)[0]
        r = r[np.lexsort((gid_loc[r], lay_loc[r]))]
        recv.append(r)
        # my owned nodes p holds: layer ON p = distance from p's owned set, measured on the mesh I was given (exact up to
        # G: every shortest path of length <= G from p's owned set to one of my owned nodes runs inside my overlap)
        lay_on_p = _node_layers(N, conn, node_owner == p, G)[old]
        s = np.where((own_loc == rank) & (lay_on_p <= G))[0]
        s = s[np.lexsort((gid_loc[s], lay_on_p[s]))]
        send.append(s)
        send_layer.append(lay_on_p[s])
    return HaloPartition(rank, world, G, X[old].copy(), conn_loc, gid_loc.copy(), lay_loc, peers, send, send_layer, recv,
                         elem_ids, src=old)


def node_owner_from_elements(n_nodes, conn, elem_owner, world):
    """Lowest rank among the owners of a node's elements (the rule of partition_from_global)."""
    first = np.full(n_nodes, world, dtype=np.int32)
    for r in range(world - 1, -1, -1):
        first[np.unique(conn[elem_owner == r])] = r
    return first


def _node_noise(gid, seed):
    """Standard-normal triples that depend on (global node id, seed) only: splitmix64 hashes -> Box-Muller."""
    def mix(z):
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    with np.errstate(over="ignore"):
        base = mix(np.asarray(gid, dtype=np.uint64) * np.uint64(6) + np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D))
        out = np.empty((len(gid), 3))
        for c in range(3):
            a = mix(base + np.uint64(2 * c + 1))
            b = mix(base + np.uint64(2 * c + 2))
            u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
            u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
            out[:, c] = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return out


def process_grid(world):
    """Blocks per axis of the weak-scaling series: every rank owns one config-sized block; 8 ranks = BASELINE config E
    (2 x 2 x 2 blocks of config C: 180 x 120 x 60 cells, 7.8 M elements); other counts: x-slabs."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world, (world, 1, 1))


def halo_block_structured(wl, config, rank, world, depth, grid=None):
    """bench.py weak scaling: rank-local construction of one config-sized block of a body of process_grid(world) blocks,
    extended by `depth` + 1 cell layers into each neighbour.  Owner of a node = the block whose half-open lattice range
    (lo, hi] holds it along every axis (the first block of an axis also owns its 0 plane); global ids = lattice index in
    the whole body, so neighbours agree on every list order without communicating.
    -> (workload dict of the extended block in the partition's local numbering, HaloPartition)"""
    cfg = wl.CONFIGS[config]
    n = np.array(cfg["cells"])
    pg = np.array(grid if grid is not None else process_grid(world))
    assert int(np.prod(pg)) == world
    b = np.array([rank % pg[0], (rank // pg[0]) % pg[1], rank // (pg[0] * pg[1])])
    ext = depth + 1
    c_lo = np.maximum(0, b * n - ext)
    c_hi = np.minimum(pg * n, (b + 1) * n + ext)
    w = wl.build(config, cells=tuple(int(v) for v in (c_hi - c_lo)), offset_cells=tuple(int(v) for v in c_lo))
    X = w["X"]
    size = np.array(cfg["size"], dtype=np.float64)
    hx = size / n / 2.0                                            # lattice spacing per axis (corner + mid-edge nodes)
    idx = np.rint(X / hx).astype(np.int64)                         # global lattice coordinates
    g = 2 * n * pg + 1
    gid = (idx[:, 2] * g[1] + idx[:, 1]) * g[0] + idx[:, 0]
    ob = np.clip((idx - 1) // (2 * n), 0, pg - 1)                  # (lo, hi] per axis
    owner = ((ob[:, 2] * pg[1] + ob[:, 1]) * pg[0] + ob[:, 0]).astype(np.int32)
    hp = halo_partition(X, w["conn"], owner, gid, rank, world, depth)
    Xl = hp.X
    # boundary conditions of the WHOLE body (test_feat10_resolution.cc:283-312): clamp at x = 0, 5000 N over x = L; every
    # rank that holds such a node carries its full value (rows are complete on every rank, nothing is shared out)
    fixed = np.where(np.abs(Xl[:, 0]) < 1e-12)[0].astype(np.int32)
    f_ext = np.zeros(3 * Xl.shape[0])
    n_face = int(g[1] * g[2])
    face = np.where(np.abs(Xl[:, 0] - size[0] * pg[0]) < 1e-9)[0]
    f_ext[3 * face] = 5000.0 / n_face
    # kernel-timing state (workloads.build): smooth field + seeded noise.  The noise must be a function of the NODE, not of
    # a rank-local index: a ghost copy whose position differs from its owner's makes the replicated rows differ
    u = 1e-2 * np.sin(np.pi * Xl / size)
    x0 = Xl + u + 1e-4 * float(hx.min()) * _node_noise(hp.l2g, 12345)
    x0[fixed] = Xl[fixed]
    wl_loc = dict(X=Xl, conn=hp.conn, fixed=fixed, f_ext=f_ext, x0=x0, material=w["material"], params=w["params"],
                  desc=w["desc"], grid=tuple(int(v) for v in pg), block=tuple(int(v) for v in b))
    return wl_loc, hp


def halo_slab_structured(wl, config, rank, world, depth):
    """x-slabs of a bar `world` times as long (the round-2 series; two ranks: identical to the block series)."""
    return halo_block_structured(wl, config, rank, world, depth, grid=(world, 1, 1))


class _DeviceBytes:
    """Zero-copy uint8 view of a raw device buffer (CUDA array interface v2)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def make_halo_exchange(torch, dist, backend):
    """-> callable(send_ptr, recv_ptr, peers, send_off, recv_off) moving byte ranges of two device buffers between
    neighbours.  'nccl': batched isend / irecv on device views (RCCL); anything else: host staging (gloo)."""
    dev = backend == "nccl"

    def ex(sp, rp, peers, so, ro):
        ns, nr = so[-1], ro[-1]
        st = torch.as_tensor(_DeviceBytes(sp, ns), device="cuda") if ns else None
        rt = torch.as_tensor(_DeviceBytes(rp, nr), device="cuda") if nr else None
        if not dev:
            st = st.cpu() if st is not None else None
            rh = torch.empty(nr, dtype=torch.uint8) if nr else None
        else:
            rh = rt
        ops = []
        for k, p in enumerate(peers):
            if so[k + 1] > so[k]:
                ops.append(dist.P2POp(dist.isend, st[so[k]:so[k + 1]], p))
            if ro[k + 1] > ro[k]:
                ops.append(dist.P2POp(dist.irecv, rh[ro[k]:ro[k + 1]], p))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if not dev and nr:
            rt.copy_(rh)
            torch.cuda.synchronize()
    return ex, (0 if dev else 1)


def attach_halo(solver, hp, torch, dist, native_rccl=None):
    """Wire a SyncedNewtonSolver to its overlapping partition: the library's own RCCL exchange when a communicator is
    given (collectives enqueued from C++ on the solver's stream and captured in its hipGraphs), else torch.distributed
    callbacks (RCCL on device views with the nccl backend, gloo through host staging in the CPU / one-GPU rehearsals)."""
    if native_rccl is not None:
        solver.SetHalo(hp, rccl_comm=native_rccl)
        return
    backend = dist.get_backend()
    ar, sync = make_allreduce(torch, dist, backend)
    ex, _ = make_halo_exchange(torch, dist, backend)
    solver.SetHalo(hp, allreduce=ar, exchange=ex, sync_before_callback=1)

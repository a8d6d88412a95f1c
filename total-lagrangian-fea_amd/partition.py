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


def rccl_communicator(dist, rank, world):
    """A RCCL communicator owned by the engine (tlfea_rccl_*): rank 0 creates the unique id, torch.distributed ships its
    128 bytes once, every rank joins.  -> opaque handle for SyncedNewtonSolver.SetInterfaceRccl / rccl_destroy."""
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
    binding.check(lib.tlfea_rccl_comm_create(box[0], int(rank), int(world), C.byref(comm)))
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

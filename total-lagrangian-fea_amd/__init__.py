"""total-lagrangian-fea_amd -- MI355X-native Total-Lagrangian element engine (T10 hot path).

Python host mirror of the reference's class surface (GPU_FEAT10_Data, SyncedNewtonSolver,
SyncedNewtonParams, ANCFCPUUtils::FEAT10_read_*, Quadrature::tet5pt_*) over the C-ABI of
`libtlfea_hip.so` (include/tlfea_c.h).  There is NO CPU fallback: creating an element-data object
without the built library or without a visible GPU raises.

Import with importlib (the directory name carries a hyphen):
    tl = importlib.import_module("total-lagrangian-fea_amd")
"""
from .binding import (LIB_PATH, TlfeaError, load_library, device_count, exported_symbols)  # noqa: F401
from .elements import GPU_ANCF3243_Data, GPU_ANCF3443_Data, GPU_FEAT10_Data  # noqa: F401
from .solvers import (SyncedNewtonParams, SyncedNewtonSolver, LinSolveOpts, SyncedAdamWNocoopParams,  # noqa: F401
                      SyncedAdamWNocoopSolver, SyncedAdamWSolver, SyncedAdamWParams, SyncedNesterovParams, SyncedNesterovSolver, SyncedVBDParams,
                      SyncedVBDSolver)
from . import mesh_utils, quadrature  # noqa: F401
from .mesh_manager import MeshManager  # noqa: F401

__all__ = ["GPU_FEAT10_Data", "GPU_ANCF3243_Data", "GPU_ANCF3443_Data", "SyncedNewtonSolver", "SyncedNewtonParams", "LinSolveOpts", "SyncedAdamWNocoopSolver", "SyncedAdamWNocoopParams", "SyncedAdamWSolver", "SyncedAdamWParams", "SyncedNesterovSolver", "SyncedNesterovParams", "SyncedVBDSolver", "SyncedVBDParams", "mesh_utils", "MeshManager",
           "quadrature", "load_library", "device_count", "TlfeaError", "LIB_PATH", "exported_symbols"]

"""ctypes binding of libtlfea_hip.so -- argument/return types for every symbol in include/tlfea_c.h."""
import ctypes as C
import importlib.util
import os
import re
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TLFEA_LIB_PATH") or os.path.join(_HERE, "libtlfea_hip.so")  # override: A/B experiments
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tlfea_c.h")
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
c_llp = C.POINTER(C.c_longlong)
HALO_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_ip, c_llp, c_llp)


class HaloListsC(C.Structure):  # tlfea_halo_lists
    _fields_ = [("n_peers", C.c_int), ("peers", c_ip), ("send_off", c_ip), ("send_nodes", c_ip), ("send_layer", c_ip),
                ("recv_off", c_ip), ("recv_nodes", c_ip), ("rank", C.c_int), ("world", C.c_int)]


class TlfeaError(RuntimeError):
    pass


class NewtonParams(C.Structure):  # SyncedNewtonParams (SyncedNewton.cuh:29-33)
    _fields_ = [("inner_atol", C.c_double), ("inner_rtol", C.c_double), ("outer_tol", C.c_double),
                ("rho", C.c_double), ("max_outer", C.c_int), ("max_inner", C.c_int), ("time_step", C.c_double)]


class AdamWParamsC(C.Structure):  # tlfea_adamw_params == SyncedAdamWParams (SyncedAdamW.cuh:27-34)
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("lr_decay", C.c_double), ("inner_tol", C.c_double),
                ("outer_tol", C.c_double), ("rho", C.c_double), ("max_outer", C.c_int), ("max_inner", C.c_int),
                ("time_step", C.c_double), ("convergence_check_interval", C.c_int), ("inner_rtol", C.c_double)]


class NesterovParamsC(C.Structure):  # tlfea_nesterov_params == SyncedNesterovParams (SyncedNesterov.cuh:26-30)
    _fields_ = [("alpha", C.c_double), ("rho", C.c_double), ("inner_tol", C.c_double), ("outer_tol", C.c_double),
                ("max_outer", C.c_int), ("max_inner", C.c_int), ("time_step", C.c_double)]


class VbdParamsC(C.Structure):  # tlfea_vbd_params == SyncedVBDParams (SyncedVBD.cuh:13-21)
    _fields_ = [("inner_tol", C.c_double), ("inner_rtol", C.c_double), ("outer_tol", C.c_double), ("rho", C.c_double),
                ("max_outer", C.c_int), ("max_inner", C.c_int), ("time_step", C.c_double), ("omega", C.c_double),
                ("hess_eps", C.c_double), ("convergence_check_interval", C.c_int), ("color_group_size", C.c_int)]


class LinSolveOptsC(C.Structure):
    _fields_ = [("rel_tol", C.c_double), ("max_iter", C.c_int), ("check_every", C.c_int), ("cheb_degree", C.c_int),
                ("cheb_kappa", C.c_double), ("cheb_bits", C.c_int), ("precond", C.c_int), ("method", C.c_int),
                ("on_unconverged", C.c_int)]


def exported_symbols():
    """Names of every function declared in include/tlfea_c.h (parsed from the header)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tlfea_[a-z0-9_]+)\s*\(", txt)) - {"tlfea_allreduce_fn"})


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and link them by UNVERSIONED file name.  If
    this library has already mapped /opt/rocm's copies (soname libamdhip64.so.7), a later `import torch` therefore maps
    the wheel's copies as a SECOND HIP runtime in the process -- and that one finds no GPU ("No HIP GPUs are
    available").  Mapping the wheel's copies first makes both sides resolve to one runtime, in either import order.
    Nothing of torch is imported or executed here; TLFEA_SYSTEM_HIP=1 keeps /opt/rocm's runtime."""
    if "torch" in sys.modules or os.environ.get("TLFEA_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # no usable bundled runtime: the system one is used


def load_library():
    """dlopen the HIP library; fails loudly when it has not been built (no CPU path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise TlfeaError(f"{LIB_PATH} is missing: build it with `make -C {_HERE}` "
                         "(or python -c 'import __graft_entry__ as g; g.build()').")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    lib.tlfea_last_error.restype = C.c_char_p
    for name in ("tlfea_t10_x12_device_ptr", "tlfea_t10_y12_device_ptr", "tlfea_t10_z12_device_ptr",
                 "tlfea_t10_external_force_device_ptr", "tlfea_t10_constraint_device_ptr",
                 "tlfea_newton_velocity_guess_device_ptr", "tlfea_adamw_velocity_guess_device_ptr", "tlfea_nesterov_velocity_guess_device_ptr",
                 "tlfea_vbd_velocity_guess_device_ptr"):
        getattr(lib, name).restype = C.c_void_p
        getattr(lib, name).argtypes = [C.c_void_p]
    _LIB = lib
    return lib


def device_count():
    return load_library().tlfea_device_count()


def check(rc):
    if rc != 0:
        raise TlfeaError(load_library().tlfea_last_error().decode())


def dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None

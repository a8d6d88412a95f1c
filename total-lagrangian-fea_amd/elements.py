"""GPU_FEAT10_Data -- host mirror of the reference class (lib_src/elements/FEAT10Data.cuh:19-852) on the
C-ABI.  Method names, argument order and call-order contract follow the reference; Eigen vectors become
NumPy arrays, Eigen::MatrixXi connectivity is an (E,10) int array (sent column-major as the reference does)."""
import ctypes as C

import numpy as np

from .binding import check, dp, ip, load_library


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GPU_FEAT10_Data:
    TYPE = "TYPE_T10"  # ElementBase.h:20
    S, Q = 10, 5       # shape functions / force quadrature points per element

    def __init__(self, num_elements, num_nodes):
        self._lib = load_library()
        self.n_elem, self.n_coef = int(num_elements), int(num_nodes)
        self._h = C.c_void_p()
        self._initialized = False

    # -- lifecycle ------------------------------------------------------------------------------
    def Initialize(self):
        check(self._lib.tlfea_t10_create(self.n_elem, self.n_coef, C.byref(self._h)))
        self._initialized = True

    def Destroy(self):
        if self._initialized:
            check(self._lib.tlfea_t10_destroy(self._h))
            self._initialized = False

    def Setup(self, tet5pt_x, tet5pt_y, tet5pt_z, tet5pt_weights, h_x12, h_y12, h_z12, element_connectivity):
        conn = np.asarray(element_connectivity)
        assert conn.shape == (self.n_elem, 10)
        conn_cm = np.ascontiguousarray(conn.T, dtype=np.int32)  # [10][E] == column-major E x 10
        qx, qy, qz, qw = _f64(tet5pt_x), _f64(tet5pt_y), _f64(tet5pt_z), _f64(tet5pt_weights)
        x, y, z = _f64(h_x12), _f64(h_y12), _f64(h_z12)
        assert x.size == y.size == z.size == self.n_coef
        check(self._lib.tlfea_t10_setup(self._h, dp(qx), dp(qy), dp(qz), dp(qw), dp(x), dp(y), dp(z), ip(conn_cm)))

    # -- setters --------------------------------------------------------------------------------
    def SetDensity(self, rho0):
        check(self._lib.tlfea_t10_set_density(self._h, C.c_double(rho0)))

    def SetDamping(self, eta_damp, lambda_damp):
        check(self._lib.tlfea_t10_set_damping(self._h, C.c_double(eta_damp), C.c_double(lambda_damp)))

    def SetSVK(self, E=None, nu=None):
        if E is None:
            check(self._lib.tlfea_t10_set_svk_select(self._h))
        else:
            check(self._lib.tlfea_t10_set_svk(self._h, C.c_double(E), C.c_double(nu)))

    def SetMooneyRivlin(self, mu10, mu01, kappa):
        check(self._lib.tlfea_t10_set_mooney_rivlin(self._h, C.c_double(mu10), C.c_double(mu01), C.c_double(kappa)))

    def SetExternalForce(self, h_f_ext):
        f = _f64(h_f_ext)
        check(self._lib.tlfea_t10_set_external_force(self._h, dp(f), int(f.size)))

    def SetNodalFixed(self, fixed_nodes):
        fx = np.ascontiguousarray(fixed_nodes, dtype=np.int32)
        check(self._lib.tlfea_t10_set_nodal_fixed(self._h, ip(fx), int(fx.size)))

    def SetLinearConstraintsCSR(self, j_offsets, j_columns, j_values, rhs):
        """General linear constraints c = J x - rhs (ANCF3243Data.cuh:810-940); columns = 3*coef + component."""
        off = np.ascontiguousarray(j_offsets, dtype=np.int32)
        col = np.ascontiguousarray(j_columns, dtype=np.int32)
        val, r = _f64(j_values), _f64(rhs)
        if off.size == 0 or off[0] != 0:
            raise ValueError("SetLinearConstraintsCSR: invalid offsets.")
        if r.size + 1 != off.size:
            raise ValueError("SetLinearConstraintsCSR: offsets/rhs size mismatch.")
        if col.size != val.size:
            raise ValueError("SetLinearConstraintsCSR: columns/values size mismatch.")
        if off[-1] != col.size:
            raise ValueError("SetLinearConstraintsCSR: offsets.back != nnz.")
        check(self._lib.tlfea_t10_set_linear_constraints_csr(self._h, int(r.size), ip(off), ip(col), dp(val), dp(r)))

    def UpdateLinearConstraintRHS(self, rhs):
        """New right-hand side of the CSR constraints, J and the sparsity stay (ANCF3443Data.cuh:977-997)."""
        r = _f64(rhs)
        check(self._lib.tlfea_t10_update_linear_constraint_rhs(self._h, dp(r), int(r.size)))

    def GetConstraintMode(self):
        return int(self._lib.tlfea_t10_get_constraint_mode(self._h))

    def UpdateNodalFixed(self, fixed_nodes):
        fx = np.ascontiguousarray(fixed_nodes, dtype=np.int32)
        check(self._lib.tlfea_t10_update_nodal_fixed(self._h, ip(fx), int(fx.size)))

    def UpdatePositions(self, h_x12, h_y12, h_z12):
        x, y, z = _f64(h_x12), _f64(h_y12), _f64(h_z12)
        check(self._lib.tlfea_t10_update_positions(self._h, dp(x), dp(y), dp(z), int(x.size)))

    def UpdateConstraintTargets(self, h_x12, h_y12, h_z12):
        x, y, z = _f64(h_x12), _f64(h_y12), _f64(h_z12)
        check(self._lib.tlfea_t10_update_constraint_targets(self._h, dp(x), dp(y), dp(z), int(x.size)))

    # -- computations -----------------------------------------------------------------------------
    def CalcDnDuPre(self):
        check(self._lib.tlfea_t10_calc_dndu_pre(self._h))

    def BuildMassCSRPattern(self):
        check(self._lib.tlfea_t10_build_mass_csr_pattern(self._h))

    def CalcMassMatrix(self):
        check(self._lib.tlfea_t10_calc_mass_matrix(self._h))

    def CalcConstraintData(self):
        check(self._lib.tlfea_t10_calc_constraint_data(self._h))

    def ConvertToCSR_ConstraintJac(self):
        check(self._lib.tlfea_t10_convert_to_csr_constraint_jac(self._h))

    def ConvertToCSR_ConstraintJacT(self):
        check(self._lib.tlfea_t10_convert_to_csr_constraint_jact(self._h))

    BuildConstraintJacobianCSR = ConvertToCSR_ConstraintJac
    BuildConstraintJacobianTransposeCSR = ConvertToCSR_ConstraintJacT

    def CalcP(self):
        check(self._lib.tlfea_t10_calc_p(self._h))

    def CalcInternalForce(self):
        check(self._lib.tlfea_t10_calc_internal_force(self._h))

    # -- getters ----------------------------------------------------------------------------------
    def get_n_elem(self):
        return self.n_elem

    def get_n_beam(self):
        return self.n_elem

    def get_n_coef(self):
        return self.n_coef

    def get_n_constraint(self):
        return self._lib.tlfea_t10_get_n_constraint(self._h)

    def Get_Is_Constraint_Setup(self):
        return bool(self._lib.tlfea_t10_is_constraint_setup(self._h))

    def GetX12DevicePtr(self):
        return self._lib.tlfea_t10_x12_device_ptr(self._h)

    def GetY12DevicePtr(self):
        return self._lib.tlfea_t10_y12_device_ptr(self._h)

    def GetZ12DevicePtr(self):
        return self._lib.tlfea_t10_z12_device_ptr(self._h)

    def GetExternalForceDevicePtr(self):
        return self._lib.tlfea_t10_external_force_device_ptr(self._h)

    def Get_Constraint_Ptr(self):
        return self._lib.tlfea_t10_constraint_device_ptr(self._h)

    # -- retrieval (reference layouts) ----------------------------------------------------------------
    def RetrieveMassCSRToCPU(self):
        nnz = C.c_int()
        check(self._lib.tlfea_t10_mass_csr_nnz(self._h, C.byref(nnz)))
        off = np.zeros(self.n_coef + 1, dtype=np.int32)
        col = np.zeros(nnz.value, dtype=np.int32)
        val = np.zeros(nnz.value)
        check(self._lib.tlfea_t10_retrieve_mass_csr(self._h, ip(off), ip(col), dp(val)))
        return off, col, val

    def RetrieveInternalForceToCPU(self):
        f = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_t10_retrieve_internal_force(self._h, dp(f)))
        return f

    def RetrieveExternalForceToCPU(self):
        f = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_t10_retrieve_external_force(self._h, dp(f)))
        return f

    def RetrievePositionToCPU(self):
        x, y, z = (np.zeros(self.n_coef) for _ in range(3))
        check(self._lib.tlfea_t10_retrieve_position(self._h, dp(x), dp(y), dp(z)))
        return x, y, z

    def RetrievePFromFToCPU(self):
        """[E][5] 3x3 matrices; flat storage is column-major per matrix like the reference."""
        P = np.zeros((self.n_elem, self.Q, 9))
        check(self._lib.tlfea_t10_retrieve_p_from_f(self._h, dp(P)))
        return P.reshape(self.n_elem, self.Q, 3, 3).transpose(0, 1, 3, 2)

    def RetrieveDeformationGradientToCPU(self):
        F = np.zeros((self.n_elem, self.Q, 9))
        check(self._lib.tlfea_t10_retrieve_deformation_gradient(self._h, dp(F)))
        return F.reshape(self.n_elem, self.Q, 3, 3).transpose(0, 1, 3, 2)

    def RetrieveDnDuPreToCPU(self):
        """[E][Q] S x 3 matrices (shape function, direction)."""
        g = np.zeros((self.n_elem, self.Q, 3, self.S))
        check(self._lib.tlfea_t10_retrieve_dndu_pre(self._h, dp(g)))
        return g.transpose(0, 1, 3, 2)

    def RetrieveDetJToCPU(self):
        d = np.zeros((self.n_elem, self.Q))
        check(self._lib.tlfea_t10_retrieve_detj(self._h, dp(d)))
        return d

    def RetrieveConnectivityToCPU(self):
        c = np.zeros((self.S, self.n_elem), dtype=np.int32)  # coefficient ids (T10: node ids)
        check(self._lib.tlfea_t10_retrieve_connectivity(self._h, ip(c)))
        return np.ascontiguousarray(c.T)

    def RetrieveConstraintDataToCPU(self):
        c = np.zeros(self.get_n_constraint())
        check(self._lib.tlfea_t10_retrieve_constraint_data(self._h, dp(c)))
        return c

    def RetrieveConstraintJacobianCSRToCPU(self):
        nc, nnz = self.get_n_constraint(), int(self._lib.tlfea_t10_constraint_jac_nnz(self._h))
        off, col, val = np.zeros(nc + 1, dtype=np.int32), np.zeros(nnz, dtype=np.int32), np.zeros(nnz)
        check(self._lib.tlfea_t10_retrieve_constraint_jac_csr(self._h, ip(off), ip(col), dp(val)))
        return off, col, val

    def RetrieveConstraintJacobianTransposeCSRToCPU(self):
        nnz = int(self._lib.tlfea_t10_constraint_jac_nnz(self._h))
        off, col, val = np.zeros(3 * self.n_coef + 1, dtype=np.int32), np.zeros(nnz, dtype=np.int32), np.zeros(nnz)
        check(self._lib.tlfea_t10_retrieve_constraint_jact_csr(self._h, ip(off), ip(col), dp(val)))
        return off, col, val

    def WriteOutputVTK(self, filename):
        check(self._lib.tlfea_t10_write_output_vtk(self._h, str(filename).encode()))


class _GPU_ANCF_Data(GPU_FEAT10_Data):
    """Common host mirror of GPU_ANCF3243_Data / GPU_ANCF3443_Data (lib_src/elements/ANCF3243Data.cuh:33-1152,
    ANCF3443Data.cuh).  4 coefficient vectors per node (r, r_u, r_v, r_w): n_coef = 4 * n_nodes;
    SetNodalFixed takes COEFFICIENT indices (ANCF3243Data.cuh:778-808)."""
    KIND = 0
    NN = 0

    def __init__(self, num_nodes, num_elements):
        self._lib = load_library()
        self.n_nodes, self.n_elem = int(num_nodes), int(num_elements)
        self.n_beam = self.n_elem
        self.n_coef = 4 * self.n_nodes
        self._h = C.c_void_p()
        self._initialized = False

    def Initialize(self):
        check(self._lib.tlfea_ancf_create(self.KIND, self.n_nodes, self.n_elem, C.byref(self._h)))
        self._initialized = True

    def PrintDsDuPre(self):
        """Text dump of the reference gradients per (element, point) (ANCF3243Data.cu:326-360)."""
        g, dj = self.RetrieveDnDuPreToCPU(), self.RetrieveDetJToCPU()
        for e in range(self.n_elem):
            for q in range(self.Q):
                print(f"\n=== Elem {e} Quadrature Point {q} detJ_ref={dj[e, q]:g} ===")
                print("        dN/dx       dN/dy       dN/dz")
                for i in range(self.S):
                    print(f"Shape {i}: " + " ".join(f"{g[e, q, i, j]:10.6f}" for j in range(3)) + " ")

    def _setup(self, length, width, height, mass_rule, force_rule, h_x12, h_y12, h_z12, connectivity):
        E = self.n_elem
        L, W, H = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (E,))) for a in
                   (length, width, height))
        mr = [_f64(a) for a in mass_rule]
        fr = [_f64(a) for a in force_rule]
        nqm = np.array([len(mr[0]), len(mr[1]), len(mr[2])], dtype=np.int32)
        nq = np.array([len(fr[0]), len(fr[1]), len(fr[2])], dtype=np.int32)
        conn = np.ascontiguousarray(np.asarray(connectivity).reshape(E, self.NN), dtype=np.int32)
        x, y, z = _f64(h_x12), _f64(h_y12), _f64(h_z12)
        assert x.size == y.size == z.size == self.n_coef
        check(self._lib.tlfea_ancf_setup(self._h, dp(L), dp(W), dp(H), dp(mr[0]), dp(mr[1]), dp(mr[2]), dp(mr[3]),
                                         dp(mr[4]), dp(mr[5]), ip(nqm), dp(fr[0]), dp(fr[1]), dp(fr[2]), dp(fr[3]),
                                         dp(fr[4]), dp(fr[5]), ip(nq), dp(x), dp(y), dp(z), ip(conn), 0))

    def CalcDsDuPre(self):
        check(self._lib.tlfea_ancf_calc_dsdu_pre(self._h))

    CalcDnDuPre = CalcDsDuPre
    RetrieveDsDuPreToCPU = GPU_FEAT10_Data.RetrieveDnDuPreToCPU

    def get_n_beam(self):
        return self.n_elem


class GPU_ANCF3243_Data(_GPU_ANCF_Data):
    TYPE, KIND, NN, S, Q = "TYPE_3243", 3243, 2, 8, 12

    def Setup(self, length, width, height, gauss_xi_m, gauss_xi, gauss_eta, gauss_zeta, weight_xi_m, weight_xi,
              weight_eta, weight_zeta, h_x12, h_y12, h_z12, h_element_connectivity):
        """Argument order of ANCF3243Data.cuh:511-521 (the mass rule shares eta/zeta with the force rule)."""
        self._setup(length, width, height,
                    (gauss_xi_m, gauss_eta, gauss_zeta, weight_xi_m, weight_eta, weight_zeta),
                    (gauss_xi, gauss_eta, gauss_zeta, weight_xi, weight_eta, weight_zeta),
                    h_x12, h_y12, h_z12, h_element_connectivity)


class GPU_ANCF3443_Data(_GPU_ANCF_Data):
    TYPE, KIND, NN, S, Q = "TYPE_3443", 3443, 4, 16, 48

    def __init__(self, num_nodes, num_elements=None):
        """(num_nodes, num_elements), or the strip constructor (num_beams): every new shell of the chain brings two
        new nodes (ANCF3443Data.cuh:445-457)."""
        if num_elements is None:
            num_elements, num_nodes = int(num_nodes), 4 + 2 * (int(num_nodes) - 1)
        super().__init__(num_nodes, num_elements)

    def Setup(self, length, width, height, gauss_xi_m, gauss_eta_m, gauss_zeta_m, gauss_xi, gauss_eta, gauss_zeta,
              weight_xi_m, weight_eta_m, weight_zeta_m, weight_xi, weight_eta, weight_zeta, h_x12, h_y12, h_z12,
              element_connectivity):
        """Argument order of ANCF3443Data.cuh:532-542."""
        self._setup(length, width, height,
                    (gauss_xi_m, gauss_eta_m, gauss_zeta_m, weight_xi_m, weight_eta_m, weight_zeta_m),
                    (gauss_xi, gauss_eta, gauss_zeta, weight_xi, weight_eta, weight_zeta),
                    h_x12, h_y12, h_z12, element_connectivity)

"""Quadrature tables of the hot path (reference: lib_utils/quadrature_utils.h:134-158)."""
import numpy as np

N_QP_T10_5 = 5
N_NODE_T10_10 = 10

# Keast 5-point rule, barycentric rows [L1,L2,L3,L4]; note the negative centroid weight.
tet5pt_bary = np.array([[0.25, 0.25, 0.25, 0.25],
                        [0.5, 1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0],
                        [1.0 / 6.0, 0.5, 1.0 / 6.0, 1.0 / 6.0],
                        [1.0 / 6.0, 1.0 / 6.0, 0.5, 1.0 / 6.0],
                        [1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0, 0.5]])
tet5pt_weights = np.array([-4.0 / 5.0, 9.0 / 20.0, 9.0 / 20.0, 9.0 / 20.0, 9.0 / 20.0]) * (1.0 / 6.0)
tet5pt_x = np.ascontiguousarray(tet5pt_bary[:, 1])
tet5pt_y = np.ascontiguousarray(tet5pt_bary[:, 2])
tet5pt_z = np.ascontiguousarray(tet5pt_bary[:, 3])

# ---- Gauss-Legendre tables used by the ANCF elements (quadrature_utils.h:8-128) --------------------------
N_SHAPE_3243, N_SHAPE_3443 = 8, 16
N_TOTAL_QP_3_2_2, N_TOTAL_QP_4_4_3 = 12, 48
gauss_xi_m_6 = np.array([-0.93246951420315202, -0.66120938646626451, -0.23861918608319691, 0.23861918608319691,
                         0.66120938646626451, 0.93246951420315202])
weight_xi_m_6 = np.array([0.17132449237917034, 0.36076157304813861, 0.46791393457269104, 0.46791393457269104,
                          0.36076157304813861, 0.17132449237917034])
gauss_xi_m_7 = np.array([-0.949107912342759, -0.741531185599394, -0.405845151377397, 0.0, 0.405845151377397,
                         0.741531185599394, 0.949107912342759])
weight_xi_m_7 = np.array([0.129484966168870, 0.279705391489277, 0.381830050505119, 0.417959183673469,
                          0.381830050505119, 0.279705391489277, 0.129484966168870])
gauss_eta_m_7, weight_eta_m_7 = gauss_xi_m_7.copy(), weight_xi_m_7.copy()
gauss_zeta_m_3 = np.array([-0.7745966692414834, 0.0, 0.7745966692414834])
weight_zeta_m_3 = np.array([0.5555555555555556, 0.8888888888888888, 0.5555555555555556])
gauss_xi_3 = np.array([-0.77459666924148340, 0.0, 0.77459666924148340])
weight_xi_3 = np.array([0.55555555555555556, 0.88888888888888889, 0.55555555555555556])
gauss_xi_4 = np.array([-0.8611363115940526, -0.3399810435848563, 0.3399810435848563, 0.8611363115940526])
weight_xi_4 = np.array([0.3478548451374538, 0.6521451548625461, 0.6521451548625461, 0.3478548451374538])
gauss_eta_2 = np.array([-0.57735026918962576, 0.57735026918962576])
weight_eta_2 = np.array([1.0, 1.0])
gauss_eta_4, weight_eta_4 = gauss_xi_4.copy(), weight_xi_4.copy()
gauss_zeta_2, weight_zeta_2 = gauss_eta_2.copy(), weight_eta_2.copy()
gauss_zeta_3, weight_zeta_3 = gauss_xi_3.copy(), weight_xi_3.copy()

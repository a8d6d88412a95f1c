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

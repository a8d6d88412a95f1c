"""MeshManager -- multi-mesh container with unified indexing (reference: lib_utils/mesh_manager.{h,cc}:60-235,180-220,
491-560).  LoadMesh / TransformMesh / TranslateMesh / GetAll* / GetMeshInstance / GetMeshIdFromElement; the NPZ
scalar-field loaders belong to the collision subsystem and are out of scope."""
from dataclasses import dataclass

import numpy as np

from .mesh_utils import FEAT10_read_elements, FEAT10_read_nodes


def rotationY(angle_rad):
    R = np.eye(4)
    c, s = np.cos(angle_rad), np.sin(angle_rad)
    R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
    return R


def translation(dx, dy, dz):
    T = np.eye(4)
    T[:3, 3] = (dx, dy, dz)
    return T


def uniformScale(s):
    S = np.eye(4)
    S[0, 0] = S[1, 1] = S[2, 2] = s
    return S


@dataclass
class MeshInstance:
    node_offset: int
    element_offset: int
    num_nodes: int
    num_elements: int
    name: str


class MeshManager:
    def __init__(self):
        self._nodes, self._elems, self._inst = [], [], []
        self._all_nodes = np.zeros((0, 3))
        self._all_elems = np.zeros((0, 0), dtype=np.int32)

    def LoadMesh(self, node_file, elem_file, name=""):
        try:
            n_nodes, nodes = FEAT10_read_nodes(node_file)
            n_elems, elems = FEAT10_read_elements(elem_file)
        except (OSError, ValueError) as exc:
            print(f"MeshManager: Failed to load mesh from {node_file} and {elem_file}: {exc}")
            return -1
        inst = MeshInstance(self.GetTotalNodes(), self.GetTotalElements(), n_nodes, n_elems,
                            name or f"mesh_{len(self._inst)}")
        self._nodes.append(nodes)
        self._elems.append(elems)
        self._inst.append(inst)
        self._rebuild()
        return len(self._inst) - 1

    def _rebuild(self):
        self._all_nodes = np.concatenate(self._nodes, axis=0)
        self._all_elems = np.concatenate([e + i.node_offset for e, i in zip(self._elems, self._inst)], axis=0).astype(np.int32)

    def TransformMesh(self, mesh_id, transform):
        inst = self.GetMeshInstance(mesh_id)
        T = np.asarray(transform, dtype=np.float64)
        X = self._nodes[mesh_id]
        hom = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1) @ T.T
        self._nodes[mesh_id] = hom[:, :3].copy()
        self._all_nodes[inst.node_offset:inst.node_offset + inst.num_nodes] = self._nodes[mesh_id]

    def TranslateMesh(self, mesh_id, dx, dy, dz):
        self.TransformMesh(mesh_id, translation(dx, dy, dz))

    def GetAllNodes(self):
        return self._all_nodes

    def GetAllElements(self):
        return self._all_elems

    def GetMeshInstance(self, mesh_id):
        if mesh_id < 0 or mesh_id >= len(self._inst):
            raise IndexError(f"MeshManager: Invalid mesh_id {mesh_id}")  # std::out_of_range in the reference
        return self._inst[mesh_id]

    def GetNumMeshes(self):
        return len(self._inst)

    def GetTotalNodes(self):
        return sum(i.num_nodes for i in self._inst)

    def GetTotalElements(self):
        return sum(i.num_elements for i in self._inst)

    def GetMeshIdFromElement(self, global_elem_idx):
        for k, i in enumerate(self._inst):
            if i.element_offset <= global_elem_idx < i.element_offset + i.num_elements:
                return k
        return -1

    def GetMeshIdFromNode(self, global_node_idx):
        for k, i in enumerate(self._inst):
            if i.node_offset <= global_node_idx < i.node_offset + i.num_nodes:
                return k
        return -1

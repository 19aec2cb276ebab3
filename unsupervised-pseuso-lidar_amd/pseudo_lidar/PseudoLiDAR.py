"""Depth image -> pseudo-LiDAR point cloud on the GPU (reference pseudo-lidar/utils/PseudoLiDAR.py; its ROS node is out of scope).

Same class surface: `PseudoLiDAR(calib_dir, sparsity)` reads `calib_velo_to_cam.txt` / `calib_cam_to_cam.txt` exactly as the
reference (:12-29, :48-67), `.T` / `.P` hold the 4x4 and 3x4 matrices, `project_PL(depth_img)` returns the [n, 4] float64 cloud.
`project_PL` accepts a [rows, cols] CUDA tensor (float32, e.g. the depth network's 1/(10 disp + 0.01) map) and returns a CUDA
tensor: un-projection, rigid transform, the x >= 0 & z < 1 m cut and the order-preserving compaction run in three small kernels
(mcav_pseudo_lidar_project); only the point count comes back to the host.  `from_matrices` builds one without calibration files.
"""
import ctypes

import numpy as np
import torch

from mcav import lib as L

L.register({
    "mcav_pseudo_lidar_workspace_bytes": (L.c_sz, [L.c_i, L.c_i]),
    "mcav_pseudo_lidar_project": (L.c_i, [L.c_p, L.c_i, L.c_i, L.c_p, L.c_p, L.c_i, L.c_p, L.c_sz, L.c_p, L.c_p, L.c_sz, L.c_p]),
})


class PseudoLiDAR:
    def __init__(self, calib_dir, sparsity):
        self.T, self.P = self.get_trans_proj(calib_dir)
        self.sparsity = sparsity

    @classmethod
    def from_matrices(cls, T, P, sparsity):
        self = object.__new__(cls)
        self.T, self.P = np.asarray(T, dtype=np.float64), np.asarray(P, dtype=np.float64)
        self.sparsity = sparsity
        return self

    def read_calib_file(self, filepath):
        """key: floats ... per line; non-float values (dates) are skipped (reference :12-29)."""
        data = {}
        with open(filepath, "r") as f:
            for line in f.readlines():
                line = line.rstrip()
                if len(line) == 0:
                    continue
                key, value = line.split(":", 1)
                try:
                    data[key] = np.array([float(x) for x in value.split()])
                except ValueError:
                    pass
        return data

    def get_trans_proj(self, calib_dir):
        velo = self.read_calib_file(calib_dir + "calib_velo_to_cam.txt")
        cam = self.read_calib_file(calib_dir + "calib_cam_to_cam.txt")
        T = np.vstack([np.concatenate((velo["R"].reshape(3, 3), velo["T"].reshape(3, 1)), axis=1), [0, 0, 0, 1]])
        return T, cam["P_rect_02"].reshape(3, 4)

    def project_PL(self, depth_img):
        depth = L.dev(torch.as_tensor(depth_img).to(torch.float32).contiguous(), "depth_img")
        if depth.dim() != 2:
            raise L.MCAVError("project_PL: depth_img must be [rows, cols]")
        rows, cols = depth.shape
        h = L.lib()
        ws = L.workspace(h.mcav_pseudo_lidar_workspace_bytes(rows, cols), depth.device, "pseudo_lidar")
        cloud = torch.empty((rows * cols, 4), dtype=torch.float64, device=depth.device)
        count = torch.zeros(1, dtype=torch.int32, device=depth.device)
        T = np.ascontiguousarray(self.T, dtype=np.float64)
        P = np.ascontiguousarray(self.P, dtype=np.float64)
        L.check(h.mcav_pseudo_lidar_project(L.ptr(depth), rows, cols, T.ctypes.data_as(ctypes.c_void_p), P.ctypes.data_as(ctypes.c_void_p),
                                            int(self.sparsity or 0), L.ptr(cloud), rows * cols, L.ptr(count), L.ptr(ws), ws.numel(), L.stream()),
                "mcav_pseudo_lidar_project")
        valid = int(count.item())
        step = int(self.sparsity) if self.sparsity else 1
        return cloud[:(valid + step - 1) // step]

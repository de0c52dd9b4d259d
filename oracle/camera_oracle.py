"""TEST INFRASTRUCTURE ONLY (not shipped, not measured): CPU restatement of the projection-matrix decomposition the reference's
DTU / BlendedMVS dataset uses, `utils/rend_util.py:31-52` -> `cv2.decomposeProjectionMatrix` (third-party: opencv-python, unpinned
in the reference's requirements.txt, NOT installed in this image).  PARITY UNPINNED against cv2 itself; what is restated is
OpenCV's published algorithm (calib3d: cvDecomposeProjectionMatrix -> cvRQDecomp3x3): three Givens rotations Qx, Qy, Qz that zero
M[2,1], M[2,0], M[1,0] in that order, then a 180-degree correction so that the first two diagonal entries of K are positive; the
camera centre is the null vector of P.  tests/test_scene_dataset.py checks neat_amd.datasets.load_K_Rt_from_P against this and
against K, R, C -> P -> K, R, C round trips."""
import numpy as np


def rq_decomp3x3_givens(M):
    M = np.asarray(M, dtype=np.float64)
    R = M.copy()
    # Qx: zero R[2,1]
    c, s = R[2, 2], R[2, 1]
    z = 1.0 / np.sqrt(c * c + s * s + 1e-300); c, s = c * z, s * z
    Qx = np.array([[1, 0, 0], [0, c, s], [0, -s, c]])
    R = R @ Qx
    # Qy: zero R[2,0]
    c, s = R[2, 2], -R[2, 0]
    z = 1.0 / np.sqrt(c * c + s * s + 1e-300); c, s = c * z, s * z
    Qy = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    R = R @ Qy
    # Qz: zero R[1,0]
    c, s = R[1, 1], R[1, 0]
    z = 1.0 / np.sqrt(c * c + s * s + 1e-300); c, s = c * z, s * z
    Qz = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
    R = R @ Qz
    Q = Qz.T @ Qy.T @ Qx.T
    # diagonal entries of R, except the last one, shall be positive: rotate by 180 degrees about an axis if necessary
    if R[0, 0] < 0:
        if R[1, 1] < 0:
            D = np.diag([-1.0, -1.0, 1.0])
        else:
            D = np.diag([-1.0, 1.0, -1.0])
        R, Q = R @ D, D @ Q
    elif R[1, 1] < 0:
        D = np.diag([1.0, -1.0, -1.0])
        R, Q = R @ D, D @ Q
    return R, Q


def load_K_Rt_from_P(P):
    P = np.asarray(P, dtype=np.float64)
    K, R = rq_decomp3x3_givens(P[:3, :3])
    _, _, vt = np.linalg.svd(P)
    t = vt[-1]                                   # null vector of P = homogeneous camera centre
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = t[:3] / t[3]
    return intrinsics, pose

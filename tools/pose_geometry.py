"""Two-view pose estimation for the accuracy harness (the pose half): what the reference's MegaDepth-1500
benchmark does with `match` + `sample` output - `estimate_pose` (romatch/utils/utils.py:30-51), `compute_pose_error`
(:126-133), `pose_auc` (:136-147), `compute_relative_pose` (:500-503).

`estimate_pose` in the reference is two OpenCV calls, `cv2.findEssentialMat(kpts0, kpts1, I, threshold, prob)` (RANSAC
around the five-point solver) and `cv2.recoverPose` (cheirality test of the four decompositions).  OpenCV is a third-party
dependency that is NOT in /root/reference and not installed here (pyproject: opencv-python, un-pinned), so this file
restates the published algorithms behind those calls and is pinned on geometry with exact answers instead of on cv2
output (tests/test_cpu_oracle.py::test_pose_*; "parity unpinned" against cv2 itself - its RANSAC is seeded by its own
RNG, so even with cv2 present parity could only be distributional):

  * five-point relative pose: D. Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004, in
    the action-matrix form of H. Stewenius, C. Engels, D. Nister, "Recent developments on direct relative orientation",
    ISPRS J. 2006: null space of the 5 x 9 epipolar system, the ten cubic constraints det(E) = 0 and
    2 E E^T E - tr(E E^T) E = 0 in (x, y, z), Gauss-Jordan elimination of the ten cubic monomials, eigenvectors of the
    10 x 10 multiplication matrix.  All samples of a RANSAC batch are solved at once (numpy, batched SVD / solve / eig).
  * RANSAC as cv2.findEssentialMat runs it: minimal samples of 5, symmetric-free Sampson distance
    (x1^T E x0)^2 / (|E x0|_{1,2}^2 + |E^T x1|_{1,2}^2) against threshold^2, iteration count adapted to the inlier ratio
    from `prob`, at most `max_iters` (cv2 default 1000).
  * recoverPose: the four (R, t) decompositions of E, triangulation of the inliers, the candidate with most points in front
    of both cameras wins (distance threshold 1e9 = the reference's argument, i.e. no far-point rejection).

Everything is numpy float64 on the host: 5 000 matches x 1 000 hypotheses is ~0.1 s, next to a 90 ms match() call the pose
step is not a GPU problem (the reference runs it on the CPU through cv2 as well).
"""
import numpy as np

# ---------------------------------------------------------------------------------------------------- small polynomial algebra
# monomials in (x, y, z) of degree <= 3, the ten cubic ones FIRST (the columns Gauss-Jordan eliminates), then the ten of the
# quotient-ring basis [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
_MONO = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
         (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_IDX = {m: i for i, m in enumerate(_MONO)}
_LIN = [_IDX[(1, 0, 0)], _IDX[(0, 1, 0)], _IDX[(0, 0, 1)], _IDX[(0, 0, 0)]]  # a degree-1 polynomial a x + b y + c z + d


def _mul_table():
    t = -np.ones((20, 20), dtype=np.int64)
    for i, a in enumerate(_MONO):
        for j, b in enumerate(_MONO):
            s = (a[0] + b[0], a[1] + b[1], a[2] + b[2])
            if s in _IDX:
                t[i, j] = _IDX[s]
    return t


_MUL = _mul_table()
_PAIRS = [(i, j, _MUL[i, j]) for i in range(20) for j in range(20) if _MUL[i, j] >= 0]


def _pmul(p, q):
    """product of polynomial batches [..., 20] x [..., 20] -> [..., 20] (terms above degree 3 never occur for our operands)"""
    out = np.zeros(np.broadcast_shapes(p.shape, q.shape), dtype=np.float64)
    pn, qn = np.any(p != 0, axis=tuple(range(p.ndim - 1))), np.any(q != 0, axis=tuple(range(q.ndim - 1)))
    for i, j, k in _PAIRS:
        if pn[i] and qn[j]:
            out[..., k] += p[..., i] * q[..., j]
    return out


def five_point(x0, x1):
    """Essential matrices of minimal samples.  x0, x1: [S, 5, 2] normalised image points (x1^T E x0 = 0).
    Returns (E [M, 3, 3], sample index [M]) - up to ten real solutions per sample."""
    S = x0.shape[0]
    a0 = np.concatenate([x0, np.ones((S, 5, 1))], axis=2)
    a1 = np.concatenate([x1, np.ones((S, 5, 1))], axis=2)
    # x1^T E x0 = sum_ij x1_i E_ij x0_j : row of the 5 x 9 system = outer(x1, x0) flattened row-major like E
    Q = (a1[:, :, :, None] * a0[:, :, None, :]).reshape(S, 5, 9)
    _, _, vt = np.linalg.svd(Q)                      # [S, 9, 9]; the last four right singular vectors span the null space
    nb = vt[:, 5:, :].reshape(S, 4, 3, 3)            # X, Y, Z, W
    # E = x X + y Y + z Z + W as a 3 x 3 matrix of degree-1 polynomials
    E = np.zeros((S, 3, 3, 20))
    for b in range(4):
        E[..., _LIN[b]] = nb[:, b]

    def matmul(Am, Bm):                               # [S, 3, 3, 20] x [S, 3, 3, 20]
        out = np.zeros((S, 3, 3, 20))
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    out[:, i, j] += _pmul(Am[:, i, k], Bm[:, k, j])
        return out

    Et = np.transpose(E, (0, 2, 1, 3))
    EEt = matmul(E, Et)
    tr = EEt[:, 0, 0] + EEt[:, 1, 1] + EEt[:, 2, 2]
    EEtE = matmul(EEt, E)
    cons = np.zeros((S, 10, 20))
    n = 0
    for i in range(3):
        for j in range(3):
            cons[:, n] = 2.0 * EEtE[:, i, j] - _pmul(tr, E[:, i, j])
            n += 1
    det = (_pmul(E[:, 0, 0], _pmul(E[:, 1, 1], E[:, 2, 2]) - _pmul(E[:, 1, 2], E[:, 2, 1]))
           - _pmul(E[:, 0, 1], _pmul(E[:, 1, 0], E[:, 2, 2]) - _pmul(E[:, 1, 2], E[:, 2, 0]))
           + _pmul(E[:, 0, 2], _pmul(E[:, 1, 0], E[:, 2, 1]) - _pmul(E[:, 1, 1], E[:, 2, 0])))
    cons[:, 9] = det
    # Gauss-Jordan on the cubic columns: cubic monomials = -B . [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
    lhs, rhs = cons[:, :, :10], cons[:, :, 10:]
    ok = np.abs(np.linalg.det(lhs)) > 1e-300
    lhs = np.where(ok[:, None, None], lhs, np.eye(10)[None])
    B = np.linalg.solve(lhs, rhs)                     # [S, 10, 10]
    # multiplication by x in the quotient ring, rows = images of the basis monomials
    A = np.zeros((S, 10, 10))
    A[:, 0] = -B[:, 0]   # x * x^2 = x^3
    A[:, 1] = -B[:, 1]   # x * xy  = x^2 y
    A[:, 2] = -B[:, 2]   # x * xz  = x^2 z
    A[:, 3] = -B[:, 3]   # x * y^2 = x y^2
    A[:, 4] = -B[:, 4]   # x * yz  = x y z
    A[:, 5] = -B[:, 5]   # x * z^2 = x z^2
    A[:, 6, 0] = 1.0     # x * x = x^2
    A[:, 7, 1] = 1.0     # x * y = xy
    A[:, 8, 2] = 1.0     # x * z = xz
    A[:, 9, 6] = 1.0     # x * 1 = x
    # A v = x v for v = the basis evaluated at a solution
    w, v = np.linalg.eig(A)
    real = (np.abs(w.imag) < 1e-9 * (1.0 + np.abs(w.real))) & ok[:, None] & (np.abs(v[:, 9, :]) > 1e-12)
    si, ei = np.nonzero(real)
    vv = v[si, :, ei].real
    xs, ys, zs = vv[:, 6] / vv[:, 9], vv[:, 7] / vv[:, 9], vv[:, 8] / vv[:, 9]
    Es = xs[:, None, None] * nb[si, 0] + ys[:, None, None] * nb[si, 1] + zs[:, None, None] * nb[si, 2] + nb[si, 3]
    nrm = np.linalg.norm(Es.reshape(-1, 9), axis=1)
    good = np.isfinite(nrm) & (nrm > 0)
    return Es[good] / nrm[good, None, None], si[good]


def sampson_sq(E, x0, x1):
    """squared Sampson distance of every correspondence to every model: E [M, 3, 3], x [N, 2] -> [M, N]"""
    a0 = np.concatenate([x0, np.ones((len(x0), 1))], axis=1)
    a1 = np.concatenate([x1, np.ones((len(x1), 1))], axis=1)
    Ex0 = np.einsum("mij,nj->mni", E, a0)
    Etx1 = np.einsum("mji,nj->mni", E, a1)
    num = np.einsum("mni,ni->mn", Ex0, a1) ** 2
    den = Ex0[..., 0] ** 2 + Ex0[..., 1] ** 2 + Etx1[..., 0] ** 2 + Etx1[..., 1] ** 2
    return num / np.maximum(den, 1e-300)


def find_essential_mat(x0, x1, threshold, prob=0.999, max_iters=1000, rng=None, batch=64):
    """cv2.findEssentialMat(x0, x1, I, RANSAC, prob, threshold) restated: returns (E [3, 3] or None, inlier mask [N] bool)."""
    rng = np.random.default_rng(0) if rng is None else rng
    n = len(x0)
    if n < 5:
        return None, np.zeros(n, dtype=bool)
    thr2 = float(threshold) ** 2
    best_E, best_mask, best_cnt = None, np.zeros(n, dtype=bool), 0
    need, done = max_iters, 0
    while done < need:
        s = min(batch, need - done)
        idx = np.stack([rng.choice(n, 5, replace=False) for _ in range(s)])
        Es, _ = five_point(x0[idx], x1[idx])
        done += s
        if len(Es) == 0:
            continue
        inl = sampson_sq(Es, x0, x1) < thr2
        cnt = inl.sum(axis=1)
        k = int(np.argmax(cnt))
        if cnt[k] > best_cnt:
            best_cnt, best_E, best_mask = int(cnt[k]), Es[k], inl[k]
            ratio = best_cnt / n
            if ratio >= 1.0:
                need = min(need, done)
            else:
                denom = np.log(max(1.0 - ratio ** 5, 1e-300))
                need = min(need, max(done, int(np.ceil(np.log(max(1.0 - prob, 1e-300)) / denom)))) if denom < 0 else need
    return best_E, best_mask


def decompose_essential(E):
    """the four (R, t) of an essential matrix (Hartley & Zisserman 9.6.2); t has unit norm"""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    R1, R2, t = U @ W @ Vt, U @ W.T @ Vt, U[:, 2]
    return [(R1, t), (R1, -t), (R2, t), (R2, -t)]


def _triangulate_depths(R, t, x0, x1):
    """linear (DLT) triangulation with P0 = [I | 0], P1 = [R | t]; returns depths in camera 0 and camera 1"""
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = np.hstack([R, t[:, None]])
    A = np.stack([x0[:, 0, None] * P0[2] - P0[0], x0[:, 1, None] * P0[2] - P0[1],
                  x1[:, 0, None] * P1[2] - P1[0], x1[:, 1, None] * P1[2] - P1[1]], axis=1)  # [N, 4, 4]
    _, _, vt = np.linalg.svd(A)
    X = vt[:, -1, :]
    X = X / np.where(np.abs(X[:, 3:]) > 1e-300, X[:, 3:], 1e-300)
    z0 = X[:, 2]
    z1 = (X[:, :3] @ R.T + t)[:, 2]
    return z0, z1


def recover_pose(E, x0, x1, mask, distance_thresh=1e9):
    """cv2.recoverPose(E, x0, x1, I, distanceThresh, mask): (number of inliers passing the cheirality test, R, t [3, 1], mask)"""
    best = (0, None, None, None)
    sel = np.nonzero(mask)[0]
    for R, t in decompose_essential(E):
        z0, z1 = _triangulate_depths(R, t, x0[sel], x1[sel])
        good = (z0 > 0) & (z1 > 0) & (z0 < distance_thresh) & (z1 < distance_thresh)
        if int(good.sum()) > best[0]:
            m = np.zeros(len(x0), dtype=bool)
            m[sel[good]] = True
            best = (int(good.sum()), R, t[:, None].copy(), m)
    return best


def estimate_pose(kpts0, kpts1, K0, K1, norm_thresh, conf=0.99999, rng=None, max_iters=1000):
    """romatch/utils/utils.py:30-51: pixel keypoints -> (R, t [3, 1], inlier mask) or None.  The normalisation is the
    reference's (inverse of K[:2, :2] and the principal point - skew included, like there)."""
    if len(kpts0) < 5:
        return None
    K0inv, K1inv = np.linalg.inv(K0[:2, :2]), np.linalg.inv(K1[:2, :2])
    x0 = (K0inv @ (kpts0 - K0[None, :2, 2]).T).T
    x1 = (K1inv @ (kpts1 - K1[None, :2, 2]).T).T
    E, mask = find_essential_mat(x0, x1, norm_thresh, prob=conf, rng=rng, max_iters=max_iters)
    if E is None:
        return None
    n, R, t, m = recover_pose(E, x0, x1, mask, 1e9)
    if n == 0:
        return None
    return R, t, mask


def compute_relative_pose(R1, t1, R2, t2):
    """utils.py:500-503"""
    rots = R2 @ R1.T
    return rots, -rots @ t1 + t2


def angle_error_mat(R1, R2):
    cos = np.clip((np.trace(R1.T @ R2) - 1) / 2, -1.0, 1.0)
    return np.rad2deg(np.abs(np.arccos(cos)))


def angle_error_vec(v1, v2):
    n = np.linalg.norm(v1) * np.linalg.norm(v2)
    return np.rad2deg(np.arccos(np.clip(np.dot(v1, v2) / n, -1.0, 1.0)))


def compute_pose_error(T_0to1, R, t):
    """utils.py:126-133: (translation direction error, rotation error) in degrees; the sign of t is not observable"""
    e_t = angle_error_vec(np.asarray(t).squeeze(), T_0to1[:3, 3])
    e_t = np.minimum(e_t, 180 - e_t)
    return e_t, angle_error_mat(R, T_0to1[:3, :3])


def pose_auc(errors, thresholds):
    """utils.py:136-147: area under the recall-vs-error curve up to each threshold, normalised"""
    errors = np.sort(np.asarray(errors, dtype=np.float64))
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors, recall = np.r_[0.0, errors], np.r_[0.0, recall]
    aucs = []
    for t in thresholds:
        last = np.searchsorted(errors, t)
        r = np.r_[recall[:last], recall[last - 1]]
        e = np.r_[errors[:last], t]
        aucs.append(float(np.sum((e[1:] - e[:-1]) * (r[1:] + r[:-1]) / 2) / t))  # np.trapz, spelled out (removed in numpy 2)
    return aucs

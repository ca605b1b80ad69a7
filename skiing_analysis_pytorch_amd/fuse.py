"""The consumers of the all-gathered [T, J, 3] joints: left/right fusion and temporal smoothing
(reference: fuse/fuse.py:93-100 softmax2, :289-325 fuse_frame_3d, :329-412 temporal_smooth_ema),
rigid right->left alignment (fuse/main_raw.py:48-93), per-joint confidences (fuse/confidence.py) and
the VideoPose3D left/right fusion without extrinsics (VideoPose3D/fuse/fuse.py, fuse_check.py).

O(T*J) host NumPy in the reference and here — the EMA is sequential in t, the data are a few KB,
and the NaN-aware masks (missing joints) must be preserved bit-for-bit, so these run on the
gathered tensor on every rank rather than as GPU kernels.  Array form ([J,3] / [T,J,3] with NaN
for missing joints) of the reference's dict form; `to_dicts` / `from_dicts` convert."""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence

import numpy as np

EPS = 1e-8


def from_dicts(seq: List[Dict[int, Iterable[float]]], target_ids: Sequence[int]) -> np.ndarray:
    """fuse.py:71-80 dict_to_array per frame -> [T, J, 3] float64, NaN where a joint is missing."""
    X = np.full((len(seq), len(target_ids), 3), np.nan, dtype=np.float64)
    for t, d in enumerate(seq):
        for j, jid in enumerate(target_ids):
            if jid in d:
                X[t, j] = np.asarray(d[jid], dtype=np.float64)
    return X


def to_dicts(X: np.ndarray, target_ids: Sequence[int]) -> List[Dict[int, np.ndarray]]:
    """fuse.py:83-89 array_to_dict per frame: only finite rows are kept."""
    return [{jid: X[t, j].copy() for j, jid in enumerate(target_ids) if np.all(np.isfinite(X[t, j]))}
            for t in range(X.shape[0])]


def softmax2(a: np.ndarray, b: np.ndarray):
    """fuse.py:93-100"""
    m = np.maximum(a, b)
    ea, eb = np.exp(a - m), np.exp(b - m)
    s = ea + eb + EPS
    return ea / s, eb / s


def _row_ok(X: np.ndarray) -> np.ndarray:
    """[..., J, 3] -> [..., J]: the joint has three finite coordinates"""
    return np.isfinite(X).all(axis=-1)


def _select_rows(code: np.ndarray, *candidates: np.ndarray) -> np.ndarray:
    """row j of the result = row j of candidates[code[j]]; code 0 = missing (NaN row)"""
    stack = np.stack((np.full_like(candidates[0], np.nan),) + candidates)          # [1 + n, J, 3]
    return np.take_along_axis(stack, code[None, :, None], axis=0)[0]


def fuse_frame_3d(Xl: np.ndarray, Xr: np.ndarray, q_l: np.ndarray, q_r: np.ndarray) -> np.ndarray:
    """fuse/fuse.py:289-325 in array form: two [J,3] estimates of the SAME frame, per-joint qualities q.
    Every joint is one of four cases, encoded as (left present) + 2 (right present):
    0 missing -> NaN, 1 -> left, 2 -> right, 3 -> softmax2-weighted blend (wl Xl + wr Xr) / (wl + wr + EPS)."""
    Xl, Xr = np.asarray(Xl, dtype=np.float64), np.asarray(Xr, dtype=np.float64)
    wl, wr = (w[:, None] for w in softmax2(q_l, q_r))
    with np.errstate(invalid="ignore"):
        blend = (wl * Xl + wr * Xr) / (wl + wr + EPS)
    return _select_rows(_row_ok(Xl).astype(np.intp) + 2 * _row_ok(Xr), Xl, Xr, blend)


# fuse/fuse.py:353-371: base smoothing factor per MHR-70 joint id, as a multiple of `alpha` -- eyes and neck
# smoother, feet and hands more responsive, everything else (limbs included) at alpha itself
_ALPHA_FACTOR = np.ones(70, dtype=np.float64)
_ALPHA_FACTOR[[1, 2, 69]] = 0.85
_ALPHA_FACTOR[[13, 14, 41, 62]] = 1.15


def temporal_smooth_ema(X: np.ndarray, target_ids: Sequence[int] = None, alpha: float = 0.7, adaptive: bool = True,
                        alpha_min: float = 0.45, alpha_max: float = 0.92, speed_gain: float = 0.25) -> np.ndarray:
    """fuse/fuse.py:329-412 on a [T, J, 3] array (NaN rows = missing joints): an exponential moving average
    run as a masked scan over time.  State = the previous output row per joint; per step a joint is
    (observed now) + 2 (has state): 0 -> NaN, 1 -> the observation starts the state, 2 -> the state is held,
    3 -> a x + (1 - a) y with a = clip(base + speed_gain |x - y|, alpha_min, alpha_max) (adaptive) or alpha."""
    X = np.asarray(X, dtype=np.float64)
    T, J = X.shape[:2]
    if T == 0:
        return X.copy()
    if adaptive:
        ids = np.arange(J) if target_ids is None else np.asarray(list(target_ids), dtype=np.int64)
        known = (ids >= 0) & (ids < _ALPHA_FACTOR.size)
        factor = np.where(known, _ALPHA_FACTOR[np.where(known, ids, 0)], 1.0)
        base = np.clip(float(alpha) * factor, alpha_min, alpha_max)
    observed = _row_ok(X)
    Y = np.empty_like(X)
    Y[0] = X[0]
    state, has_state = Y[0], observed[0]
    for t in range(1, T):
        x = X[t]
        with np.errstate(invalid="ignore"):
            if adaptive:
                a = np.clip(base + speed_gain * np.linalg.norm(x - state, axis=1), alpha_min, alpha_max)[:, None]
            else:
                a = float(alpha)
            mix = a * x + (1.0 - a) * state
        Y[t] = _select_rows(observed[t].astype(np.intp) + 2 * has_state, x, state, mix)
        state, has_state = Y[t], _row_ok(Y[t])
    return Y


# ---------------------------------------------------------------------------------------------
# Left / right alignment and confidences (the rest of SURVEY f2).  Arrays use NaN rows for missing
# joints where the reference uses dicts without the key; `from_dicts` / `to_dicts` convert.
# ---------------------------------------------------------------------------------------------
def _finite_rows(X: np.ndarray) -> np.ndarray:
    return np.isfinite(X).all(axis=1)


def kabsch_rigid_align(src: np.ndarray, dst: np.ndarray):
    """fuse/main_raw.py:48-68.  Rotation R (det +1) and translation t with R @ src_i + t ~ dst_i
    (no scale): SVD of the cross-covariance of the centred sets, last right-singular vector
    flipped when the product would be a reflection."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    cs, cd = src.mean(axis=0), dst.mean(axis=0)
    U, _, Vt = np.linalg.svd((src - cs).T @ (dst - cd))
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[-1, :] *= -1
        R = Vt.T @ U.T
    return R, cd - R @ cs


def align_right_to_left(X_left: np.ndarray, X_right: np.ndarray) -> np.ndarray:
    """fuse/main_raw.py:71-93.  Right-view joints [J, 3] moved into the left view's frame by the rigid
    transform fitted on the joints present in both; fewer than 3 common joints -> unchanged; joints
    missing on either side keep their right-view value."""
    X_left = np.asarray(X_left, dtype=np.float64)
    X_right = np.asarray(X_right, dtype=np.float64)
    both = _finite_rows(X_left) & _finite_rows(X_right)
    if int(both.sum()) < 3:
        return X_right.copy()
    R, t = kabsch_rigid_align(X_right[both], X_left[both])
    out = X_right.copy()
    out[both] = X_right[both] @ R.T + t
    return out


def fit_weakpersp_3d_to_2d(X3d: np.ndarray, U2d: np.ndarray, min_points: int = 8):
    """fuse/confidence.py:9-60.  u ~ s * (X @ M) + t with M (3, 2) of orthonormal columns, fitted on the
    rows finite in both inputs.  -> (s, M, t, mask of the rows used)."""
    X3d = np.asarray(X3d, dtype=np.float64)
    U2d = np.asarray(U2d, dtype=np.float64)
    if X3d.ndim != 2 or X3d.shape[1] != 3 or U2d.shape != (X3d.shape[0], 2):
        raise AssertionError("fit_weakpersp_3d_to_2d: X3d (N, 3) and U2d (N, 2) expected")
    used = _finite_rows(X3d) & _finite_rows(U2d)
    if int(used.sum()) < min_points:
        raise ValueError(f"Not enough valid points to fit: {int(used.sum())} < {min_points}")
    X, U = X3d[used], U2d[used]
    mx, mu = X.mean(axis=0, keepdims=True), U.mean(axis=0, keepdims=True)
    Xc, Uc = X - mx, U - mu
    P, S, Qt = np.linalg.svd(Xc.T @ Uc, full_matrices=True)
    M = P[:, :2] @ Qt
    energy = (Xc ** 2).sum()
    if energy < 1e-12:
        raise ValueError("Degenerate 3D points (too small variance).")
    s = S.sum() / energy
    t = (mu - s * (mx @ M)).reshape(2)
    return float(s), M, t, used


def weakpersp_reproj_confidence(X3d: np.ndarray, U2d: np.ndarray, sigma_px: float = 12.0, min_points: int = 8,
                                eps: float = 1e-12):
    """fuse/confidence.py:63-103.  Per-joint confidence exp(-e^2 / (2 sigma^2)) of the weak-perspective
    reprojection residual e (pixels); 0 where the residual is undefined.
    -> (conf [N], err [N], Uhat [N, 2], dict(s, M, t, valid_used))."""
    X3d = np.asarray(X3d, dtype=np.float64)
    U2d = np.asarray(U2d, dtype=np.float64)
    s, M, t, used = fit_weakpersp_3d_to_2d(X3d, U2d, min_points=min_points)
    Uhat = s * (X3d @ M) + t
    err = np.full(X3d.shape[0], np.nan)
    ok = _finite_rows(U2d) & _finite_rows(Uhat)
    diff = Uhat[ok] - U2d[ok]
    err[ok] = np.sqrt((diff ** 2).sum(axis=1))
    var = max(float(sigma_px), eps) ** 2
    conf = np.zeros_like(err)
    fin = np.isfinite(err)
    conf[fin] = np.exp(-(err[fin] ** 2) / (2.0 * var))
    return conf, err, Uhat, {"s": s, "M": M, "t": t, "valid_used": used}


def _unit(v: np.ndarray, eps: float) -> np.ndarray:
    n = np.linalg.norm(v)
    return v * 0.0 if n < eps else v / n


def canonicalize_pose_3d(X: np.ndarray, root_idx: int, left_hip_idx: int, right_hip_idx: int, left_shoulder_idx: int,
                         right_shoulder_idx: int, scale_mode: str = "hip", eps: float = 1e-9):
    """fuse/confidence.py:112-176.  Body-centred frame: origin at the root joint, x = left hip -> right
    hip, y = hip centre -> shoulder centre made orthogonal to x, z = x cross y; divided by the hip
    width ("hip") or the hip-shoulder distance ("torso").  Any key joint missing or a degenerate
    scale -> all-NaN pose, NaN rotation, NaN scale.  -> (Xc [N, 3], R (world -> canonical), s)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[1] != 3:
        raise AssertionError("canonicalize_pose_3d: X (N, 3) expected")
    nothing = (np.full_like(X, np.nan), np.full((3, 3), np.nan), np.nan)
    keys = [root_idx, left_hip_idx, right_hip_idx, left_shoulder_idx, right_shoulder_idx]
    if not np.isfinite(X[keys]).all():
        return nothing
    X0 = X - X[root_idx]
    lh, rh = X0[left_hip_idx], X0[right_hip_idx]
    hips = 0.5 * (lh + rh)
    shoulders = 0.5 * (X0[left_shoulder_idx] + X0[right_shoulder_idx])
    ex = _unit(rh - lh, eps)
    ey = _unit(shoulders - hips, eps)
    ez = _unit(np.cross(ex, ey), eps)
    ey = _unit(np.cross(ez, ex), eps)
    R = np.stack([ex, ey, ez], axis=0)
    if scale_mode == "hip":
        s = np.linalg.norm(rh - lh)
    elif scale_mode == "torso":
        s = np.linalg.norm(shoulders - hips)
    else:
        raise ValueError("scale_mode must be 'hip' or 'torso'")
    if not np.isfinite(s) or s < eps:
        return nothing
    return (R @ X0.T).T / s, R, float(s)


def crossview_consistency_confidence(X_a: np.ndarray, X_b: np.ndarray, *, root_idx: int, left_hip_idx: int,
                                     right_hip_idx: int, left_shoulder_idx: int, right_shoulder_idx: int,
                                     sigma_3d: float = 0.08, scale_mode: str = "hip", eps: float = 1e-12):
    """fuse/confidence.py:178-224.  Per-joint confidence exp(-d^2 / (2 sigma^2)) of the distance d between
    the two views' poses after each is put into its own body-centred frame.
    -> (conf [N], dist [N], Xa_c, Xb_c, dict(Ra, sa, Rb, sb, valid))."""
    X_a = np.asarray(X_a, dtype=np.float64)
    X_b = np.asarray(X_b, dtype=np.float64)
    kw = dict(root_idx=root_idx, left_hip_idx=left_hip_idx, right_hip_idx=right_hip_idx,
              left_shoulder_idx=left_shoulder_idx, right_shoulder_idx=right_shoulder_idx, scale_mode=scale_mode)
    A, Ra, sa = canonicalize_pose_3d(X_a, **kw)
    B, Rb, sb = canonicalize_pose_3d(X_b, **kw)
    dist = np.full(X_a.shape[0], np.nan)
    ok = _finite_rows(A) & _finite_rows(B)
    diff = A[ok] - B[ok]
    dist[ok] = np.sqrt((diff ** 2).sum(axis=1))
    var = max(float(sigma_3d), eps) ** 2
    conf = np.zeros_like(dist)
    fin = np.isfinite(dist)
    conf[fin] = np.exp(-(dist[fin] ** 2) / (2.0 * var))
    return conf, dist, A, B, {"Ra": Ra, "sa": sa, "Rb": Rb, "sb": sb, "valid": ok}


# ---- VideoPose3D left / right fusion without extrinsics (VideoPose3D/fuse/fuse.py, fuse_check.py) ----
H36M_HIP, H36M_NECK, H36M_L_HIP, H36M_R_HIP, H36M_L_SHO, H36M_R_SHO = 0, 9, 4, 1, 11, 14
H36M_TORSO = [H36M_HIP, H36M_NECK, H36M_L_HIP, H36M_R_HIP, H36M_L_SHO, H36M_R_SHO]


def center_scale_h36m(X: np.ndarray):
    """VideoPose3D/fuse/fuse.py:19-28: pelvis at the origin, pelvis-neck distance 1 (left alone if < 1e-8)."""
    pelvis, neck = X[H36M_HIP].copy(), X[H36M_NECK].copy()
    s = np.linalg.norm(neck - pelvis)
    s = s if s > 1e-8 else 1.0
    return (X - pelvis) / s, s


def estimate_rigid_umeyama(X: np.ndarray, Y: np.ndarray, allow_scale: bool = False):
    """VideoPose3D/fuse/fuse_check.py:24-71.  R, t, s meant to give s * R @ Y_i + t ~ X_i, on the rows
    finite in both.  Quirk kept for parity: with Sigma = Yc^T Xc / N = U S Vt the reference takes
    R = U @ Vt, the transpose of the least-squares rotation V @ U^T."""
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y, dtype=float)
    ok = _finite_rows(X) & _finite_rows(Y)
    X, Y = X[ok], Y[ok]
    n = X.shape[0]
    if n < 3:
        raise ValueError("at least 3 non-collinear correspondences are needed")
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    Sigma = (Yc.T @ Xc) / n
    U, S, Vt = np.linalg.svd(Sigma)
    R = U @ Vt
    reflected = False
    if np.linalg.det(R) < 0:
        U[:, -1] *= -1
        R = U @ Vt
        reflected = True
    s = S.sum() / ((Yc ** 2).sum() / n + 1e-12) if allow_scale else 1.0
    t = mx - s * (R @ my)
    info = {"num_points": n, "singular_values": S, "reflect_fixed": reflected,
            "Sigma_rank": int(np.linalg.matrix_rank(Sigma)), "cond_Sigma": (S[0] / S[-1]) if S[-1] > 0 else np.inf}
    return R, t, s, info


def fuse_two(L: np.ndarray, R_aligned: np.ndarray, tau=0.08, wL=None, wR=None) -> np.ndarray:
    """VideoPose3D/fuse/fuse.py:57-98.  Per joint: the only finite side, else the higher-weight side when
    the two disagree by more than tau[j], else the weighted mean; NaN when both are missing."""
    J = L.shape[0]
    tau = np.full(J, float(tau)) if isinstance(tau, (float, int)) else np.asarray(tau, dtype=float)
    wL = np.ones(J) if wL is None else np.asarray(wL, dtype=float)
    wR = np.ones(J) if wR is None else np.asarray(wR, dtype=float)
    lok, rok = _finite_rows(L), _finite_rows(R_aligned)
    out = np.full_like(L, np.nan)
    only_l, only_r, both = lok & ~rok, rok & ~lok, lok & rok
    out[only_l] = L[only_l]
    out[only_r] = R_aligned[only_r]
    for j in np.nonzero(both)[0]:
        if np.linalg.norm(L[j] - R_aligned[j]) > tau[j]:
            out[j] = L[j] if wL[j] >= wR[j] else R_aligned[j]
        else:
            out[j] = (wL[j] * L[j] + wR[j] * R_aligned[j]) / (wL[j] + wR[j] + 1e-9)
    return out


def fuse_pose_no_extrinsics_h36m(left_3d: np.ndarray, right_3d: np.ndarray, tau: float = 0.08, allow_scale: bool = False,
                                 mirror_right_x: bool = False, wL=None, wR=None, return_diagnostics: bool = True):
    """VideoPose3D/fuse/fuse.py:101-215 (without its console output).  H36M-17 poses [17, 3] or [T, 17, 3]
    of the two views -> one pose per frame in the pelvis-origin, pelvis-neck-normalised frame of the left
    view: normalise both, fit the right torso onto the left torso (estimate_rigid_umeyama), fuse per
    joint, normalise again.  Diagnostics as the reference: per-frame dicts, mean_gain, bad_frames."""
    left_3d = np.asarray(left_3d)
    L, R = left_3d, np.asarray(right_3d)
    single = L.ndim == 2
    if single:
        L = L[None]
    if R.ndim == 2:
        R = R[None]
    if L.shape != R.shape or L.shape[1:] != (17, 3):
        raise AssertionError("inputs must both be (*, 17, 3)")
    T = L.shape[0]

    def weights(w):
        if w is None:
            return np.ones((T, 17), dtype=float)
        w = np.asarray(w)
        return np.tile(w[None, :], (T, 1)) if w.ndim == 1 else w

    wl, wr = weights(wL), weights(wR)
    fused_seq = np.empty_like(L)
    diag = {"per_frame": [], "mean_gain": None, "bad_frames": []} if return_diagnostics else None
    for t in range(T):
        Rt = R[t].copy()
        if mirror_right_x:
            Rt[:, 0] *= -1
            Rt[:, 2] *= -1
        Ln, _ = center_scale_h36m(L[t].copy())
        Rn, _ = center_scale_h36m(Rt)
        Rm, tv, s, _ = estimate_rigid_umeyama(Ln[H36M_TORSO], Rn[H36M_TORSO], allow_scale=allow_scale)
        fused, _ = center_scale_h36m(fuse_two(Ln, s * (Rm @ Rn.T).T + tv, tau=tau, wL=wl[t], wR=wr[t]))
        fused_seq[t] = fused
        if return_diagnostics:
            before = float(np.linalg.norm(Ln - Rn, axis=-1).mean())
            to_l = float(np.linalg.norm(fused - Ln, axis=-1).mean())
            to_r = float(np.linalg.norm(fused - Rn, axis=-1).mean())
            gain = before - 0.5 * (to_l + to_r)
            diag["per_frame"].append({"frame": t, "LR_before": before, "Fused_vs_L": to_l, "Fused_vs_R": to_r, "gain": gain,
                                      "s": s, "R": Rm, "t": tv})
            if gain < 0:
                diag["bad_frames"].append(t)
    if return_diagnostics:
        gains = [d["gain"] for d in diag["per_frame"]]
        diag["mean_gain"] = float(np.nanmean(gains)) if gains else np.nan
    return (fused_seq[0] if single else fused_seq), diag

"""The consumers of the all-gathered [T, J, 3] joints: left/right fusion and temporal smoothing
(reference: fuse/fuse.py:93-100 softmax2, :289-325 fuse_frame_3d, :329-412 temporal_smooth_ema).

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


def fuse_frame_3d(Xl: np.ndarray, Xr: np.ndarray, q_l: np.ndarray, q_r: np.ndarray) -> np.ndarray:
    """fuse.py:289-325: per-joint softmax-weighted fusion of two [J,3] estimates in the SAME frame;
    a joint finite on one side only is taken from that side; missing on both stays NaN."""
    Xl, Xr = np.asarray(Xl, dtype=np.float64), np.asarray(Xr, dtype=np.float64)
    ok_l = np.all(np.isfinite(Xl), axis=1)
    ok_r = np.all(np.isfinite(Xr), axis=1)
    wl, wr = softmax2(q_l, q_r)
    fused = np.full_like(Xl, np.nan)
    both = ok_l & ok_r
    fused[both] = (wl[both, None] * Xl[both] + wr[both, None] * Xr[both]) / (wl[both, None] + wr[both, None] + EPS)
    only_l = ok_l & ~ok_r
    fused[only_l] = Xl[only_l]
    only_r = ok_r & ~ok_l
    fused[only_r] = Xr[only_r]
    return fused


def temporal_smooth_ema(X: np.ndarray, target_ids: Sequence[int] = None, alpha: float = 0.7, adaptive: bool = True,
                        alpha_min: float = 0.45, alpha_max: float = 0.92, speed_gain: float = 0.25) -> np.ndarray:
    """fuse.py:329-412 on a [T, J, 3] array (NaN = missing): EMA with per-joint base alpha
    (MHR-70 ids: eyes+neck smoother, feet+hands more responsive) and speed-adaptive alpha."""
    X = np.asarray(X, dtype=np.float64)
    T, J = X.shape[:2]
    if T == 0:
        return X.copy()
    if target_ids is None:
        target_ids = list(range(J))
    core_ids, limb_ids, endpoint_ids = {1, 2, 69}, {5, 6, 7, 8, 9, 10, 11, 12}, {13, 14, 41, 62}
    alpha_joint = np.full((J,), float(alpha), dtype=np.float64)
    if adaptive:
        for j, jid in enumerate(target_ids):
            if jid in core_ids:
                alpha_joint[j] = alpha * 0.85
            elif jid in limb_ids:
                alpha_joint[j] = alpha * 1.00
            elif jid in endpoint_ids:
                alpha_joint[j] = alpha * 1.15
        alpha_joint = np.clip(alpha_joint, alpha_min, alpha_max)
    Y = np.full_like(X, np.nan)
    Y[0] = X[0]
    for t in range(1, T):
        xt, yp = X[t], Y[t - 1]
        ok_x = np.all(np.isfinite(xt), axis=1)
        ok_p = np.all(np.isfinite(yp), axis=1)
        both = ok_x & ok_p
        if np.any(both):
            if adaptive:
                speed = np.linalg.norm(xt[both] - yp[both], axis=1)
                a = np.clip(alpha_joint[both] + speed_gain * speed, alpha_min, alpha_max)
            else:
                a = np.full((np.count_nonzero(both),), float(alpha), dtype=np.float64)
            Y[t, both] = a[:, None] * xt[both] + (1.0 - a)[:, None] * yp[both]
        miss_x = ~ok_x & ok_p
        Y[t, miss_x] = yp[miss_x]
        miss_p = ok_x & ~ok_p
        Y[t, miss_p] = xt[miss_p]
    return Y

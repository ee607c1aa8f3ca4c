"""CPU restatement (numpy, float32) of the reference's adaptive coarse-graining of observed Hi-C matrices -
TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/make_golden.py, never by the product path).

Follows /root/reference/selene_utils2.py:274-463 (`adaptive_coarsegrain_gpu`) and its non-square wrapper
:466-504 (`_adaptive_coarsegrain`), statement by statement, including what the code does rather than what its
comments say:
  * the replacement mask uses the min of the RAW counts of the 2x2 block with INVALID pixels counted as 0
    (`countar_next_mask` is built at :437-438 and never used; :440 coarsens `countar_next`), so every block that
    contains an invalid pixel is replaced at every level;
  * values are float32 throughout (`torch.empty(..., dtype=torch.float)`, :398), sums are (row pair) then (column pair)
    (:349-351), the valid-pixel counts are integers;
  * `val_cur = ar_cur / armask_cur` is NaN where a coarse pixel has no valid fine pixel; such fine pixels are
    invalid themselves and are zeroed (:444) and finally set to NaN (:451).
Pinned by tests/golden/G18_coarsegrain.npz, generated from the reference function itself (tools/make_golden.py
--coarsegrain patches `np.int` and the CUDA default tensor type in the generator process only)."""
import numpy as np


def _coarsen_sum(a):
    m = a.shape[0] // 2
    r = a.reshape(m, 2, m, 2)
    s = r[:, 0] + r[:, 1]              # axis 1 (rows of the block)
    return s[:, :, 0] + s[:, :, 1]     # then axis 2 (columns)


def _coarsen_min(a):
    m = a.shape[0] // 2
    r = np.nan_to_num(a.reshape(m, 2, m, 2), nan=np.inf)
    return r.min(axis=1).min(axis=2)


def _expand(a):
    return np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)


def adaptive_coarsegrain(ar, countar, cutoff=5, max_levels=8, min_shape=8):
    """selene_utils2.py:274-463 for a SQUARE matrix; returns float32 [n, n]."""
    ar = np.asarray(ar)
    countar = np.asarray(countar)
    norig = ar.shape[0]
    nlog = np.log2(norig)
    if not np.allclose(nlog, np.rint(nlog)):
        newn = int(2 ** np.ceil(nlog))
        a = np.full((newn, newn), np.nan, dtype=np.float32)
        c = np.zeros((newn, newn), dtype=np.float32)
        a[:norig, :norig] = ar
        c[:norig, :norig] = countar
    else:
        a, c = ar.astype(np.float32).copy(), countar.astype(np.float32).copy()
    mask = np.isfinite(a)
    c[~mask] = 0
    a[~mask] = 0
    ars, cnts, masks = [a], [c], [mask.astype(np.int64)]
    for _ in range(max_levels):
        if cnts[-1].shape[0] > min_shape:
            cnts.append(_coarsen_sum(cnts[-1]))
            masks.append(_coarsen_sum(masks[-1]))
            ars.append(_coarsen_sum(ars[-1]))
    ar_cur, mask_cur = ars.pop(), masks.pop()
    cnts.pop()
    ar_next, mask_next = ar_cur, mask_cur
    for _ in range(len(cnts)):
        ar_next, cnt_next, mask_next = ars.pop(), cnts.pop(), masks.pop()
        with np.errstate(invalid="ignore", divide="ignore"):
            val_cur = ar_cur / mask_cur.astype(np.float32)
        addar = _expand(val_cur) * mask_next.astype(np.float32)
        cur = _expand(_coarsen_min(cnt_next)) < cutoff
        ar_next[cur] = addar[cur]
        ar_next[mask_next == 0] = 0
        ar_cur, mask_cur = ar_next, mask_next
    ar_next = ar_next.copy()
    ar_next[mask_next == 0] = np.nan
    return ar_next[:norig, :norig]


def adaptive_coarsegrain_any_shape(ar, countar, max_levels=12):
    """selene_utils2.py:466-504: tiny and non-square inputs are padded with NaN to a square first."""
    ar, countar = np.asarray(ar), np.asarray(countar)
    assert ar.shape == countar.shape
    h, w = ar.shape
    if h < 9 and w < 9:
        a = np.full((9, 9), np.nan)
        c = np.full((9, 9), np.nan)
        a[:h, :w], c[:h, :w] = ar, countar
        return adaptive_coarsegrain(a, c, max_levels=max_levels)[:h, :w]
    if h == w:
        return adaptive_coarsegrain(ar, countar, max_levels=max_levels)
    n = max(h, w)
    a = np.full((n, n), np.nan)
    c = np.full((n, n), np.nan)
    a[:h, :w], c[:h, :w] = ar, countar
    return adaptive_coarsegrain(a, c, max_levels=max_levels)[:h, :w]

"""The few OpenCV calls on the prediction path, restated in numpy (cv2 is not a dependency).

cv2.resize(img, (W//f, H//f)) with the default INTER_LINEAR (predict.py:378-381),
cv2.dilate / cv2.erode with a ones(k,k) kernel, default anchor and border
(predict.py:428,437; noise.py:20,26).
"""
import numpy as np


_KINDS = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def _downsample_native(img: np.ndarray, f: int):
    """The even-factor case in the library (rope_downsample_even: the same four taps and roundings in one pass, host
    code, interpreter lock released); None when the library is not built or the layout is not row-contiguous."""
    kind = _KINDS.get(img.dtype)
    ch = 1 if img.ndim == 2 else img.shape[2]
    if kind is None or img.ndim not in (2, 3) or img.strides[1] != ch * img.itemsize or (img.ndim == 3 and img.strides[2] != img.itemsize) \
            or img.strides[0] < 0:
        return None
    try:
        import ctypes as C
        from .engine import load_library
        lib = load_library()
    except Exception:                                    # noqa: BLE001 — host preparation only: the numpy path below is the same arithmetic
        return None
    H, W = img.shape[:2]
    out = np.empty((H // f, W // f) + img.shape[2:], img.dtype)
    rc = lib.rope_downsample_even(C.c_void_p(img.ctypes.data), H, W, ch, img.strides[0], f, kind, out.ctypes.data_as(C.c_void_p))
    return out if rc == 0 else None


def resize_linear(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """Bilinear resize with OpenCV's pixel-centre alignment: src = (dst+0.5)*scale-0.5, edge
    clamped.  Float images use float weights in two passes (rows of x first, then y);
    uint8 images use OpenCV's 11-bit fixed-point weights with round-half-up."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    if (out_w, out_h) == (W, H):
        return img.copy()
    if W % out_w == 0 and H % out_h == 0 and W // out_w == H // out_h and (W // out_w) % 2 == 0:
        # even integer factor f: both taps have weight 1/2 and sit at f/2-1, f/2 of every block, so the four
        # strided views below are exactly the gathers of the general path (same operations, same order)
        f = W // out_w
        native = _downsample_native(img, f)
        if native is not None:
            return native
        a, b = f // 2 - 1, f // 2
        # gather the four taps first, convert after: only 4/f^2 of the frame is ever touched
        taa, tab, tba, tbb = img[a::f, a::f], img[a::f, b::f], img[b::f, a::f], img[b::f, b::f]
        if img.dtype == np.uint8:
            taa, tab, tba, tbb = (t.astype(np.int64) for t in (taa, tab, tba, tbb))
            r0 = (taa * 1024 + tab * 1024) >> 4
            r1 = (tba * 1024 + tbb * 1024) >> 4
            return np.clip((((1024 * r0) >> 16) + ((1024 * r1) >> 16) + 2) >> 2, 0, 255).astype(np.uint8)
        wt = np.float64 if img.dtype == np.float64 else np.float32
        taa, tab, tba, tbb = (t.astype(wt) for t in (taa, tab, tba, tbb))
        h = wt(0.5)
        top = taa * h + tab * h
        bot = tba * h + tbb * h
        return (top * h + bot * h).astype(img.dtype)

    def taps(n_out, n_in):
        scale = n_in / n_out
        src = (np.arange(n_out) + 0.5) * scale - 0.5
        i0 = np.floor(src).astype(np.int64)
        w1 = (src - i0).astype(np.float32)
        lo = i0 < 0
        i0c = np.clip(i0, 0, n_in - 1)
        i1c = np.clip(i0 + 1, 0, n_in - 1)
        w1 = np.where(lo, 0.0, w1).astype(np.float32)
        w1 = np.where(i0 >= n_in - 1, 0.0, w1).astype(np.float32)
        return i0c, i1c, (1.0 - w1).astype(np.float32), w1

    x0, x1, ax0, ax1 = taps(out_w, W)
    y0, y1, ay0, ay1 = taps(out_h, H)
    if img.dtype == np.uint8:
        S = 2048                                           # INTER_RESIZE_COEF_SCALE
        cx0, cx1 = np.rint(ax0 * S).astype(np.int64), np.rint(ax1 * S).astype(np.int64)
        cy0, cy1 = np.rint(ay0 * S).astype(np.int64), np.rint(ay1 * S).astype(np.int64)
        src = img.astype(np.int64)
        shp = (1, -1) + (1,) * (img.ndim - 2)
        rows = src[:, x0] * cx0.reshape(shp) + src[:, x1] * cx1.reshape(shp)
        shp_y = (-1, 1) + (1,) * (img.ndim - 2)
        # OpenCV: ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2
        r0, r1 = rows[y0] >> 4, rows[y1] >> 4
        out = (((cy0.reshape(shp_y) * r0) >> 16) + ((cy1.reshape(shp_y) * r1) >> 16) + 2) >> 2
        return np.clip(out, 0, 255).astype(np.uint8)
    work = img.astype(np.float64 if img.dtype == np.float64 else np.float32)
    shp = (1, -1) + (1,) * (img.ndim - 2)
    rows = work[:, x0] * ax0.reshape(shp).astype(work.dtype) + work[:, x1] * ax1.reshape(shp).astype(work.dtype)
    shp_y = (-1, 1) + (1,) * (img.ndim - 2)
    out = rows[y0] * ay0.reshape(shp_y).astype(work.dtype) + rows[y1] * ay1.reshape(shp_y).astype(work.dtype)
    return out.astype(img.dtype)


def _window_reduce(img: np.ndarray, k: int, fn, pad_value) -> np.ndarray:
    """k x k sliding reduce with anchor at k//2 (OpenCV's default for even kernels too)."""
    a = k // 2
    before, after = a, k - 1 - a
    H, W = img.shape
    p = np.full((H + k - 1, W + k - 1), pad_value, dtype=img.dtype)
    p[before:before + H, before:before + W] = img
    out = p[0:H, 0:W].copy()
    for dy in range(k):
        for dx in range(k):
            if dy or dx:
                out = fn(out, p[dy:dy + H, dx:dx + W])
    del after
    return out


def dilate(img: np.ndarray, k: int) -> np.ndarray:
    """cv2.dilate(img, ones((k,k))): max over the window, borders ignored (padded with -inf)."""
    img = np.asarray(img)
    pad = -np.inf if img.dtype.kind == 'f' else np.iinfo(img.dtype).min
    return _window_reduce(img, k, np.maximum, pad)


def erode(img: np.ndarray, k: int) -> np.ndarray:
    """cv2.erode(img, ones((k,k))): min over the window, borders ignored (padded with +inf)."""
    img = np.asarray(img)
    pad = np.inf if img.dtype.kind == 'f' else np.iinfo(img.dtype).max
    return _window_reduce(img, k, np.minimum, pad)

"""Headless preview of the matching process: the four-quadrant frame of the reference's ProjectionViz
(robotpose/prediction/predict.py:510-602) — input colour blended with its depth, detected links, the candidate
render, render depth minus input depth — composed with numpy and appended to an uncompressed AVI instead of an
OpenCV window (cv2 is not in this image and a GPU box has no display).

A display aid, not part of the parity contract: the frame layout, blend weights, nearest/linear resizes and the
`out == tgt` grey rule follow the reference; the colour map is a polynomial fit of Turbo rather than OpenCV's
table, and the four captions are not drawn (no font rasteriser here)."""
import struct

import numpy as np

from ..constants import VIDEO_FPS
from ..imgproc import resize_linear


def _turbo_lut() -> np.ndarray:
    """256x3 uint8 BGR table; 5th-order polynomial approximation of the Turbo map in each channel."""
    x = np.arange(256) / 255.0
    p = np.stack([x ** k for k in range(6)], 1)
    r = p @ [0.13572138, 4.61539260, -42.66032258, 132.13108234, -152.94239396, 59.28637943]
    g = p @ [0.09140261, 2.19418839, 4.84296658, -14.18503333, 4.27729857, 2.82956604]
    b = p @ [0.10667330, 12.64194608, -60.58204836, 110.36276771, -89.90310912, 27.34824973]
    return (np.clip(np.stack([b, g, r], 1), 0, 1) * 255 + .5).astype(np.uint8)


_TURBO = _turbo_lut()


def color_array(x: np.ndarray, mn: float = None, mx: float = None, percent: float = 3, ignore_zero: bool = True) -> np.ndarray:
    """Scale to 0..255 between the percent-th / (100-percent)-th percentiles and colour-map (utils.py:185-226)."""
    x = np.asarray(x, float)
    zero = x == 0
    if mn is None:
        nz = x[~zero]
        mn = (np.percentile(nz, percent) if nz.size else 0.0) if ignore_zero else np.min(x)
    if mx is None:
        mx = np.percentile(x, 100 - percent) if ignore_zero else np.max(x)
    span = (mx - mn) if mx != mn else 1.0
    out = _TURBO[np.clip((x - mn) / span * 255, 0, 255).astype(np.uint8)]
    if ignore_zero:
        out[zero] = (0, 0, 0)
    return out


def resize_nearest(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.INTER_NEAREST: source index floor(dst * scale)."""
    h, w = img.shape[:2]
    r = np.minimum((np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    c = np.minimum((np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return img[r][:, c]


class AviWriter:
    """Uncompressed BGR24 AVI (RIFF 'AVI ' with one 'vids'/DIB stream).  Every frame is one '00db' chunk,
    rows bottom-up as DIBs are; sizes and the frame count are patched into the headers on close()."""

    def __init__(self, path: str, fps: int, resolution):
        self.w, self.h = int(resolution[0]), int(resolution[1])
        self.fps, self.frames = int(fps), 0
        self.row = (self.w * 3 + 3) & ~3
        self.f = open(path, 'wb')
        self._header()

    def _header(self):
        frame_bytes = self.row * self.h
        strf = struct.pack('<IiiHHIIiiII', 40, self.w, self.h, 1, 24, 0, frame_bytes, 0, 0, 0, 0)
        strh = struct.pack('<4s4sIHHIIIIIIIIhhhh', b'vids', b'DIB ', 0, 0, 0, 0, 1, self.fps, 0, self.frames,
                           frame_bytes, 0xFFFFFFFF, 0, 0, 0, self.w, self.h)
        strl = b'strl' + b'strh' + struct.pack('<I', len(strh)) + strh + b'strf' + struct.pack('<I', len(strf)) + strf
        avih = struct.pack('<IIIIIIIIIIIIII', 1000000 // self.fps, frame_bytes * self.fps, 0, 0x10, self.frames, 0, 1,
                           frame_bytes, self.w, self.h, 0, 0, 0, 0)
        hdrl = b'hdrl' + b'avih' + struct.pack('<I', len(avih)) + avih + b'LIST' + struct.pack('<I', len(strl)) + strl
        movi_bytes = 4 + self.frames * (8 + frame_bytes)
        body = b'AVI ' + b'LIST' + struct.pack('<I', len(hdrl)) + hdrl + b'LIST' + struct.pack('<I', movi_bytes) + b'movi'
        self.f.seek(0)
        self.f.write(b'RIFF' + struct.pack('<I', len(body) + movi_bytes - 4) + body)

    def write(self, frame: np.ndarray):
        assert frame.shape == (self.h, self.w, 3) and frame.dtype == np.uint8
        rows = np.zeros((self.h, self.row), np.uint8)
        rows[:, :self.w * 3] = frame[::-1].reshape(self.h, self.w * 3)
        self.f.seek(0, 2)
        self.f.write(b'00db' + struct.pack('<I', rows.size))
        self.f.write(rows.tobytes())
        self.frames += 1

    def release(self):
        if self.f is not None:
            self._header()
            self.f.close()
            self.f = None


class ProjectionViz:
    """Same surface as the reference's: load*() the inputs of a frame and each candidate render, show() composes
    the frame (kept in .frame, counted in .shown) and appends it to the video when a path was given."""

    def __init__(self, video_path: str = None, fps: int = VIDEO_FPS, resolution=(1280, 720)):
        self.write_to_file = video_path is not None
        self.resolution = resolution
        self.writer = AviWriter(video_path, fps, resolution) if self.write_to_file else None
        self.res = np.flip(np.array(self.resolution))
        self.resize_to = tuple(int(v) for v in np.array(self.resolution) // 2)
        self.frame = np.zeros((*self.res, 3), dtype=np.uint8)
        self.input_side_up_to_date = False
        self.shown = 0

    def loadTargetColor(self, target_color: np.ndarray) -> None:
        self.tgt_color = target_color
        self.input_side_up_to_date = False

    def loadTargetDepth(self, target_depth: np.ndarray) -> None:
        self.tgt_depth = target_depth
        self.input_side_up_to_date = False

    def loadSegmentedLinks(self, segmented_color: np.ndarray) -> None:
        self.seg_links = segmented_color
        self.input_side_up_to_date = False

    def loadRenderedColor(self, render_color: np.ndarray) -> None:
        self.rend_color = render_color

    def loadRenderedDepth(self, render_depth: np.ndarray) -> None:
        self.rend_depth = render_depth

    def _genInput(self):
        hh, hw = self.res[0] // 2, self.res[1] // 2
        self.frame[:hh, :hw] = self._orig()
        self.frame[hh:, :hw] = self._seg()
        self.input_side_up_to_date = True

    def show(self) -> None:
        if not self.input_side_up_to_date:
            self._genInput()
        hh, hw = self.res[0] // 2, self.res[1] // 2
        self.frame[:hh, hw:] = resize_linear(self.rend_color, *self.resize_to)
        self.frame[hh:, hw:] = self._depth()
        self.frame[hh - 1:hh + 2, :] = 255                      # cv2.line(..., thickness=3)
        self.frame[:, hw - 1:hw + 2] = 255
        self.shown += 1
        if self.write_to_file:
            self.writer.write(self.frame)

    def _seg(self):
        return resize_linear(self.seg_links, *self.resize_to)

    def _orig(self):
        COLOR_ALPHA = .6
        color = resize_linear(self.tgt_color, *self.resize_to).astype(float)
        depth = color_array(resize_linear(np.asarray(self.tgt_depth, float), *self.resize_to), percent=5).astype(float)
        return np.clip(np.rint(color * COLOR_ALPHA + depth * (1 - COLOR_ALPHA)), 0, 255).astype(np.uint8)

    def _depth(self):
        tgt_d = resize_nearest(np.asarray(self.tgt_depth, float), *self.resize_to)
        d = resize_nearest(np.asarray(self.rend_depth, float), *self.resize_to)
        out = tgt_d - d
        out[out == tgt_d] = 0                                   # nothing rendered here: no difference to show
        colored = color_array(out)
        colored[out == tgt_d] = (55, 55, 55)                    # evaluated AFTER the zeroing, as the reference does: both empty
        return colored

    def close(self):
        if self.writer is not None:
            self.writer.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModellessProjectionViz(ProjectionViz):
    """The modelless camera predictor's preview (camera_pose_prediction.py:502-575): the same four quadrants without the
    detected links (that quadrant stays black)."""

    def _genInput(self):
        hh, hw = self.res[0] // 2, self.res[1] // 2
        self.frame[:hh, :hw] = self._orig()
        self.input_side_up_to_date = True


"""Segmenters for `Predictor(segmenter=...)`.

The reference runs a PixelLib/Matterport Mask R-CNN (`custom_segmentation.segmentImage`,
robotpose/prediction/predict.py:94-98,416) and reads `class_ids`, `scores` and `masks (H,W,K)` from
its result.  Any callable with that output shape can stand in; the trained network itself is the
next item of SURVEY §8f (no weights are available offline).

`ColorSegmenter` serves colour-coded frames (synthetic renders, auto-annotated data): one instance
per link whose flat render colour is present, score 1.0 — the same information the reference's
synthetic mode reads directly (predict.py:445-469), delivered through the segmentation path so that
`_segmentLoad` (instance merge, dilate 8 / erode 7 body mask, in-place depth masking) is exercised.
"""
import numpy as np

from .constants import DEFAULT_RENDER_COLORS


class ColorSegmenter:

    stateless = True            # frames may be handed over from several threads and in any order (Predictor.run_many's workers)

    def __init__(self, class_names, color_dict=None, split_instances: bool = False):
        """class_names: ["BG", link names...] as Predictor.classes; color_dict: name -> [b,g,r] (defaults to
        DEFAULT_RENDER_COLORS in link order); split_instances: emit two half-masks per link, to exercise the
        merge of several instances of one class (predict.py:383-395)."""
        self.class_names = list(class_names)
        self.colors = color_dict or {n: DEFAULT_RENDER_COLORS[i] for i, n in enumerate(self.class_names[1:])}
        self.split = split_instances

    def __call__(self, color: np.ndarray) -> dict:
        ids, scores, masks = [], [], []
        for cid, name in enumerate(self.class_names):
            if cid == 0 or name not in self.colors:
                continue
            c = self.colors[name]
            m = (color[..., 0] == c[0]) & (color[..., 2] == c[2])      # blue alone cannot tell base_link from background
            if not m.any():
                continue
            parts = [m]
            if self.split:
                cols = np.arange(m.shape[1])[None, :] < np.median(np.where(m)[1])
                parts = [p for p in (m & cols, m & ~cols) if p.any()]
            for k, p in enumerate(parts):
                ids.append(cid)
                scores.append(1.0 - 0.1 * k)
                masks.append(p)
        if not masks:
            return {'class_ids': np.zeros(0, int), 'scores': np.zeros(0), 'masks': np.zeros(color.shape[:2] + (0,), bool)}
        return {'class_ids': np.array(ids), 'scores': np.array(scores), 'masks': np.stack(masks, -1)}

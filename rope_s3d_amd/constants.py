"""Numeric knobs of the prediction path.

Values mirror the reference's module constants (robotpose/constants.py:11-32,60-91)
so that crops, lookup grids and render colours come out identical.
"""
import numpy as np

MAX_LINKS = 7                     # constants.py:11
NUM_RENDER_LINKS = 6              # link_6_t is never rendered (render_utils.py:31-32)

# Crop search (constants.py:19-23)
CROP_RENDER_WEIGHTING = (6, 3, 3, 0, 1, 0)
CROP_VARYING = 'SLUB'
CROP_MAX_PER_JOINT = 50
CROP_SEC_ALLOTTED_APPROX = 20
CROP_PADDING = 10

# Lookup (constants.py:28-32)
GPU_MEMORY_ALLOWED_FOR_LOOKUP = 0.1
LOOKUP_MAX_DIV_PER_LINK = 200
LOOKUP_JOINTS = 'SLU'
LOOKUP_NUM_RENDERED = 6

DEFAULT_CAMERA_POSE = [0, -1.5, .75, 0, 0, 0]      # constants.py:60
RENDERER_FALLBACK_CAMERA_POSE = [0.04, -1.425, 0.75, 0, -0.02, -0.05]   # render.py:49

JOINT_LETTERS = 'SLURBT'          # utils.py:54

# pyrender.IntrinsicsCamera defaults used by the reference (projection.py:161-169)
ZNEAR = 0.05
ZFAR = 100.0


def render_colors(num: int = MAX_LINKS):
    """Flat segmentation colours, channel 0 unique per link (constants.py:65-91).

    b = linspace(0,255,num) truncated to int, g = 0, r = |255 - 2b|.
    """
    b = np.linspace(0, 255, num).astype(int)
    return [[int(bi), 0, int(abs(255 - 2 * bi))] for bi in b]


DEFAULT_RENDER_COLORS = render_colors(MAX_LINKS)
# channel-0 value per rendered link id; 255 marks background in the engine's id image
LINK_BLUE = np.array([c[0] for c in DEFAULT_RENDER_COLORS], dtype=np.uint8)
BACKGROUND_ID = 255
VIDEO_FPS = 15      # default preview video frames per second (constants.py:58)
JSON_LINK_FILE = r"\\marvin\ROPE\joint_states.json"      # where the controller drops its joint state (constants.py:16)

#!/usr/bin/env python3
"""One Predictor alone and beside the network running flat out on a second thread, for several interpreter switch intervals
(profiles/r02_seg_boxes.txt section 4 also lists the variants with the network on a CU-masked stream and beside threads that
only hold the interpreter lock or only launch tiny kernels: edit the loop at the end)."""
import ctypes, os, sys, threading, time
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
from rope_s3d_amd.maskrcnn import MaskRCNNSegmenter

def hip_lib():
    for line in open('/proc/self/maps'):
        if 'libamdhip64' in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError('no libamdhip64 mapped')

def masked_stream(keep_of_4):
    hip = hip_lib()
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if i % 4 < keep_of_4:
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

seg = MaskRCNNSegmenter(7, device='cuda:0', seed=0, min_confidence=0.0)
frames = [np.random.default_rng(i).integers(0, 255, (90, 160, 3), dtype=np.uint8) for i in range(8)]
for _ in range(3): seg.batch(frames)

from rope_s3d_amd import SyntheticPredictor
from rope_s3d_amd.constants import DEFAULT_CAMERA_POSE
sp = SyntheticPredictor(DEFAULT_CAMERA_POSE, '1280_720_color', 8, 'SLU', noise=False, seed=1)
p = sp.predictor
lim = sp.urdf_reader.joint_limits
poses = [np.random.default_rng(7919 + f).uniform(lim[:, 0], lim[:, 1]) * np.array([1, 1, 1, 0, 0, 0]) for f in range(200)]
sp.run(poses[0])
fr = []
for q in poses:
    sp.renderer.setJointAngles(q); fr.append(sp.renderer.render())

def predictor_rate():
    t = time.perf_counter()
    for f in fr: p.run(*f)
    return len(fr) / (time.perf_counter() - t)

print('predictor alone: %.0f fps' % predictor_rate())
for sw in (5e-3, 1e-3, 2e-4, 5e-5, 1e-5):
  sys.setswitchinterval(sw)
  for label, stream in (('switch interval %g: network on a plain stream' % sw, torch.cuda.Stream()),):
      stop, count = [False], [0]
      def work():
          with torch.cuda.stream(stream):
              while not stop[0]:
                  for _ in seg.batches([frames] * 4): count[0] += 8
      th = threading.Thread(target=work); th.start()
      time.sleep(0.3); c0, t0 = count[0], time.perf_counter()
      r = predictor_rate()
      segrate = (count[0] - c0) / (time.perf_counter() - t0)
      stop[0] = True; th.join()
      print(f'{label}: predictor {r:.0f} fps, network {segrate:.0f} fps')

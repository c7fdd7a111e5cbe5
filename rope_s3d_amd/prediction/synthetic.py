"""Synthetic self-consistency harness (reference: robotpose/prediction/synthetic.py:11-75).

Render the robot at a known pose, feed colour+depth to the Predictor in synthetic mode
(link masks read from the colour image), compare predicted with actual.
"""
import numpy as np

from ..simulation.noise import NoiseMaker
from ..simulation.render import Renderer
from ..urdf import URDFReader
from ..utils import str_to_arr
from .predict import Predictor


class SyntheticPredictor:

    def __init__(self, camera_pose, base_intrin, ds_factor, do_angles, noise, *, device: int = 0, seed: int = None,
                 lookup_divisions=None):
        self.renderer = Renderer(camera_pose=camera_pose, camera_intrin=base_intrin, device=device)
        self.predictor = Predictor(camera_pose, ds_factor, do_angles=do_angles, base_intrin=base_intrin,
                                   color_dict=self.renderer.color_dict, device=device,
                                   lookup_divisions=lookup_divisions)
        self.urdf_reader = URDFReader()
        self.do_angles = do_angles
        self.rng = np.random.default_rng(seed)
        self.noise = NoiseMaker(self.rng)
        self.do_noise = noise

    def run(self, pose=None):
        if pose is None:
            pose = self._generatePose()
        self.renderer.setJointAngles(pose)
        color, depth = self.renderer.render()
        if self.do_noise:
            depth = self.noise.holes(depth)
        predicted = self.predictor.run(color, depth)
        return pose, predicted

    def _generatePose(self):
        lim = self.urdf_reader.joint_limits
        return self.rng.uniform(lim[:, 0], lim[:, 1]) * str_to_arr(self.do_angles)

    def run_batch(self, number: int, file: str = 'synth_test'):
        return self.run_batch_poses([None] * number, file)

    def run_batch_poses(self, poses, file: str = 'synth_test'):
        if not file.endswith('.npy'):
            file += '.npy'
        results = np.zeros((2, len(poses), 6))            # [actual, predicted]
        for i in range(len(poses)):
            results[0, i], results[1, i] = self.run(poses[i])
            if i % 250 == 0:                               # periodic partial save (synthetic.py:57-58)
                np.save(file, results)
        np.save(file, results)
        return results

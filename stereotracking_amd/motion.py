"""Constant-velocity Kalman filter over (cx, cy, aspect, h) used by the depth-guided association
step.  Host-side (numpy fp64) by design: north_star keeps the tracker on the CPU.

Behavioural spec: reference mmtrack/models/motion/kalman_filter.py:38-189 (8-state filter,
position std = h/20, velocity std = h/160, aspect stds 1e-2 / 1e-5 / 1e-1, Cholesky update).
"""
import numpy as np
import scipy.linalg

from .registry import TASK_UTILS


@TASK_UTILS.register_module()
class KalmanFilter:
    chi2inv95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070, 6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}

    def __init__(self, center_only=False, use_nsa=False):
        self.center_only = center_only
        self.gating_threshold = self.chi2inv95[2 if center_only else 4]
        self.use_nsa = use_nsa
        self._F = np.eye(8)          # state transition, dt = 1
        self._F[:4, 4:] = np.eye(4)
        self._H = np.eye(4, 8)       # measurement matrix
        self._w_pos = 1.0 / 20
        self._w_vel = 1.0 / 160

    def _stds(self, h, k_pos, k_vel, aspect_pos, aspect_vel):
        p, v = k_pos * self._w_pos * h, k_vel * self._w_vel * h
        return [p, p, aspect_pos, p, v, v, aspect_vel, v]

    def initiate(self, measurement):
        """(x, y, a, h) -> mean (8,), covariance (8, 8) of a new track (zero velocity)."""
        mean = np.r_[measurement, np.zeros_like(measurement)]
        std = self._stds(measurement[3], 2, 10, 1e-2, 1e-5)
        return mean, np.diag(np.square(std))

    def predict(self, mean, covariance):
        std = self._stds(mean[3], 1, 1, 1e-2, 1e-5)
        motion_cov = np.diag(np.square(np.r_[std[:4], std[4:]]))
        new_mean = np.dot(self._F, mean)
        new_cov = np.linalg.multi_dot((self._F, covariance, self._F.T)) + motion_cov
        return new_mean, new_cov

    def project(self, mean, covariance, bbox_score=0.):
        p = self._w_pos * mean[3]
        std = [p, p, 1e-1, p]
        if self.use_nsa:
            std = [(1 - bbox_score) * x for x in std]
        innovation_cov = np.diag(np.square(std))
        return np.dot(self._H, mean), np.linalg.multi_dot((self._H, covariance, self._H.T)) + innovation_cov

    def update(self, mean, covariance, measurement, bbox_score=0.):
        proj_mean, proj_cov = self.project(mean, covariance, bbox_score)
        chol, lower = scipy.linalg.cho_factor(proj_cov, lower=True, check_finite=False)
        gain = scipy.linalg.cho_solve((chol, lower), np.dot(covariance, self._H.T).T, check_finite=False).T
        new_mean = mean + np.dot(measurement - proj_mean, gain.T)
        new_cov = covariance - np.linalg.multi_dot((gain, proj_cov, gain.T))
        return new_mean, new_cov

"""Shared builders for tests (synthetic observations follow SURVEY.md §8(d))."""
import numpy as np


def synth_observation(rng, C=2, G=10):
  lidar = (rng.integers(0, 6, size=(200, 200, C)) / 5.0) * (rng.random((200, 200, C)) < 0.12)
  goal = np.cumsum(np.abs(rng.normal(size=(G, 2))) * 2.0, axis=0)
  goal = np.c_[goal, np.zeros((G, 1))]
  return dict(
      lidar=lidar.astype(np.float32),
      velocity=rng.normal(0, 3.0, size=(3,)).astype(np.float32),
      is_at_traffic_light=np.float32(rng.random() < 0.2),
      traffic_light_state=np.float32(rng.integers(0, 4)),
      goal=goal.astype(np.float32),
  )

"""oatomobile_amd — MI355X-native deep-imitative-model inference path of OATML/oatomobile.

Public surface mirrors `oatomobile.baselines.torch` (baselines/torch/__init__.py:17-21) for the
path in scope: `ImitativeModel`, `RIPAgent`, `DIMAgent`.
"""

from oatomobile_amd.agents import DIMAgent
from oatomobile_amd.agents import RIPAgent
from oatomobile_amd.agents import SetPointAgent
from oatomobile_amd.model import ImitativeModel
from oatomobile_amd.model import transform_visual

from oatomobile_amd.cil import BehaviouralModel
from oatomobile_amd.cil import CILAgent
from oatomobile_amd.lidar import lidar_to_bev
from oatomobile_amd.train import DIMTrainer

__all__ = ["ImitativeModel", "RIPAgent", "DIMAgent", "SetPointAgent", "transform_visual", "lidar_to_bev",
           "BehaviouralModel", "CILAgent", "DIMTrainer"]

"""Same import path shape as the reference: `from oatomobile_amd.baselines.torch import ImitativeModel, RIPAgent`
(cf. oatomobile/baselines/torch/__init__.py:17-21)."""
from oatomobile_amd.agents import DIMAgent
from oatomobile_amd.agents import RIPAgent
from oatomobile_amd.model import ImitativeModel

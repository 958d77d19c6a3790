"""Operator layer mirroring the reference's ``libs/model`` package
(libs/model/__init__.py:1-2 pre-imports the heat-map backbones so that the
``eval('models.heatmapModel.<name>.get_pose_net')`` plugin lookup of
libs/model/egonet.py:43-44 resolves)."""
from . import heatmapModel  # noqa: F401
from .heatmapModel import hrnet  # noqa: F401

from models.backbones.resnet import *  # noqa: F401,F403  (registry = this module's __dict__, src/algorithms/base.py:34-37)

"""Prediction-side model classes (mirror of stardist/models for predict_instances*)."""
from .config import Config2D, Config3D
from .model2d import StarDist2D
from .model3d import StarDist3D
from . import pretrained

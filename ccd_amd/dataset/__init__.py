"""Data pipeline of the pretraining step: LMDB records -> the batch contract of
Dino/dataset/datasetsupervised_kmeans.py:82-87."""
from .datasetsupervised_kmeans import DeviceViewMaker, ImageDatasetSelfSupervisedKmeans, collate_uint8  # noqa: F401

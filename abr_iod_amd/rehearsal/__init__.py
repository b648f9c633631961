"""Rehearsal-buffer construction ("prototype box selection"): the offline pass that picks, per new class,
the ground-truth boxes whose RoI feature maps are most typical and writes their crops as the replay memory."""
from .extract_memory import Mem  # noqa: F401
from .prototype_box_selection import extract_bboxes_and_features, selector  # noqa: F401

"""Evaluation helpers around the reconstruction: alignment of the spline trajectory with ground truth."""

"""Inference algorithms.  Only the CAVIaR path is in scope (reference circuitmap/optimise/__init__.py:1-4)."""
from .caviar import caviar  # noqa: F401

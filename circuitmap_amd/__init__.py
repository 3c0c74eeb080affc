"""circuitmap_amd -- MI355X-native CAVIaR: drop-in for circuitmap's `Model.fit(method='caviar')` path.

Public names follow the reference package (circuitmap/__init__.py:1-3): `Model`, `NeuralDemixer`, `simulate`.
Imports are lazy (the reference's eager import pulls torch + pytorch_lightning + jax) so that importing the package never
needs a GPU.
"""

__all__ = ["Model", "NeuralDemixer", "simulate", "optimise"]


def __getattr__(name):
    if name == "Model":
        from .model import Model
        return Model
    if name == "NeuralDemixer":
        from .neural_waveform_demixing import NeuralDemixer
        return NeuralDemixer
    if name == "simulate":
        from .simulation import simulate
        return simulate
    if name == "optimise":
        import importlib
        return importlib.import_module(".optimise", __name__)
    raise AttributeError("module 'circuitmap_amd' has no attribute '%s'" % name)

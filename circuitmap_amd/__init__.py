"""circuitmap_amd -- MI355X-native CAVIaR: drop-in for circuitmap's `Model.fit(method='caviar')` path.

Public names follow the reference package (circuitmap/__init__.py:1-3): `Model`, `simulate`.
`NeuralDemixer` (the PSC demixing network) is outside this hot path and is not provided.
Imports are lazy so that importing the package never needs a GPU.
"""

__all__ = ["Model", "simulate", "optimise"]


def __getattr__(name):
    if name == "Model":
        from .model import Model
        return Model
    if name == "simulate":
        from .simulation import simulate
        return simulate
    if name == "optimise":
        import importlib
        return importlib.import_module(".optimise", __name__)
    raise AttributeError("module 'circuitmap_amd' has no attribute '%s'" % name)

"""circuitmap_amd -- MI355X-native CAVIaR (drop-in for circuitmap's `Model.fit(method='caviar')` path)."""

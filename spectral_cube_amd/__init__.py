"""MI355X-native per-spaxel reduction engine for spectral cubes."""
__version__ = "0.1.0"

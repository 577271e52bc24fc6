from .grid import GridEncoder, grid_encode  # noqa: F401  (reference gridencoder/__init__.py:1 exports GridEncoder)

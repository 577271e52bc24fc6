from .raymarching import *  # noqa: F401,F403  (same star-export as the reference's raymarching/__init__.py:1)

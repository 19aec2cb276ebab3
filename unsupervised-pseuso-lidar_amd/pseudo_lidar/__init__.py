from .PseudoLiDAR import PseudoLiDAR  # noqa: F401

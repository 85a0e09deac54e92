"""StabNet hot path on MI355X (gfx950): ResNet-v2-50 regressor + multi-grid warp behind the
reference's call surface.  Import as `stabnet_amd` (see ../stabnet_amd/__init__.py)."""
__version__ = "0.1.0"

"""cvcs_amd: MI355X-native per-tile segmentation hot path behind the reference's factory surface
(load_network / load_loss / load_optimizer, source/scripts/utils.py:174-242 of theElandor/CVCS)."""
__version__ = "0.1.0"

"""MI355X-native PointNet++ set-abstraction / feature-propagation path."""
__version__ = "0.1.0"

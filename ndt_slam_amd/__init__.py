"""MI355X-native NDT scan-matching core behind the reference's PoseEstimator boundary."""
__version__ = "0.1.0"

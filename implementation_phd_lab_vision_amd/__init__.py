"""MI355X-native ResNet-50 feature extraction (hot path of the reference's
``src/preprocess_resnet_features.py``).  See DESIGN.md."""
__version__ = "0.1.0"

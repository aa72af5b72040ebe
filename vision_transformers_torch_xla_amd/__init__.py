"""MI355X-native ViT training path: timm-style create_model() / engine.train_one_epoch() over libvitk (HIP, gfx950)."""
__version__ = "0.1.0"

"""`from models import build_model` of the reference (models/__init__.py:13-14)."""
from .conditional_detr import build


def build_model(args):
    return build(args)

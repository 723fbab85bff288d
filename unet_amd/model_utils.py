"""Factories with the reference's names and signatures (models/model_utils.py:5-41)."""
import torch

from .unet import UNet


def create_unet(in_channels=3, out_channels=1):
    """models/model_utils.py:5"""
    return UNet(in_channels, out_channels)


def create_lunext(in_channels=3, out_channels=1):
    """models/model_utils.py:19"""
    from .lunext import LUNeXt
    return LUNeXt(in_channels, out_channels)


def save_model(model, filepath):
    """models/model_utils.py:33 -- plain state_dict checkpoint, interchangeable with the reference's."""
    torch.save(model.state_dict(), filepath)
    print(f"model saved to: {filepath}")

"""unet_amd: MI355X (gfx950) implementation of the SaviorDT/UNet training/inference hot path.

Public surface (same names and call conventions as the reference):
    UNet, LUNeXt                      models/unet.py, models/lunext.py
    soft_dice_cldice, soft_dice       losses/cl_dice.py
    create_unet, create_lunext, save_model   models/model_utils.py
    train_model                       trainer/training_utils.py
    evaluate_model, calculate_metrics trainer/k_fold.py
    save_results                      test_helper/utils.py
    DeviceDataset                     the device-resident counterpart of training_utils.py:52-84 (TensorDataset + DataLoader)
Everything executes in hand-written HIP kernels behind the C ABI of include/unet_hip.h.
"""
from ._lib import lib, LIB_PATH, HipKernelError          # noqa: F401  (fails loudly if the .so is missing)
from .unet import UNet                                   # noqa: F401
from .lunext import LUNeXt                               # noqa: F401
from .model_utils import create_unet, create_lunext, save_model   # noqa: F401
from .cl_dice import soft_dice_cldice, soft_cldice, soft_dice, SoftSkeletonize   # noqa: F401
from .self_reg import SelfRegLoss                         # noqa: F401
from .data import DeviceDataset                         # noqa: F401
from .trainer import train_model, evaluate_model, calculate_metrics, save_results   # noqa: F401

__all__ = ["train_model", "evaluate_model", "calculate_metrics", "save_results", "UNet", "LUNeXt", "soft_dice_cldice", "soft_cldice", "soft_dice", "SoftSkeletonize", "create_unet", "create_lunext", "save_model", "DeviceDataset", "lib", "LIB_PATH", "HipKernelError"]

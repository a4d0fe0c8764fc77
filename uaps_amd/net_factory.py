"""`net_factory` with the reference's signature and behaviour (utilities/UAPS_net_factory.py:5-13):
known names give a module placed on the current ROCm device (the reference calls `.cuda()`),
unknown names give None.  On a machine without a GPU the module stays on the CPU instead of raising,
so checkpoints can be inspected anywhere; the loss/perturbation kernels still require the GPU."""
import torch

from .res_uaps import ResUAPS
from .unet import UNet, UNet_UAPS


def net_factory(net_type="unet_uaps", in_chns=3, class_num=4, n_aux=3):
    if net_type == "unet":
        net = UNet(in_chns=in_chns, class_num=class_num)
    elif net_type == "unet_uaps":
        net = UNet_UAPS(in_chns=in_chns, class_num=class_num, n_aux=n_aux)
    elif net_type in ("resnet50_uaps", "resnet101_uaps", "resnet152_uaps"):     # not in the reference: BASELINE.json configs[4]
        net = ResUAPS(in_chns=in_chns, class_num=class_num, n_aux=n_aux, backbone=net_type.split("_")[0])
    else:
        return None
    return net.cuda() if torch.cuda.is_available() else net

"""uaps_amd: MI355X-native training step of UAPS (uncertainty-aware, dynamically mixed pseudo-labels).

Public surface (mirrors the names the reference's UAPS_train.py imports):
  net_factory, UNet_UAPS, UNet                      (utilities/UAPS_net_factory.py, UAPS_unet.py)
  dice_loss, ce_loss, uaps_sup_loss, uaps_unsup_loss, uaps_step_loss   (UAPS_train.py:186-282)
  sigmoid_rampup, get_current_consistency_weight    (utilities/ramps.py, UAPS_train.py:81-87)
  FeatureNoise, Dropout, FeatureDropout             (utilities/UAPS_unet.py:156-185)
  mIoU, mDice, pixel_accuracy, seg_confusion        (utilities/metrics.py)
  softmax_mse_loss, softmax_kl_loss, kl_loss, entropy_map, entropy_minmization   (utilities/losses_1.py, losses_2.py)
  uncertainty_map                                   (UAPS-Testing.ipynb cell 24)
  sibling.cct_consistency_loss / ucc_pseudo_supervision / uamt_consistency_loss   (CCT_train.py:195-199,
                                                    UCC_train.py:213-238, UA_MT_train.py:188-214)
  UAPSTrainer                                       (UAPS_train.py:109-450 step/optimizer/checkpoint)
"""
from .ramps import sigmoid_rampup, get_current_consistency_weight
from .losses import (dice_loss, ce_loss, uaps_sup_loss, uaps_unsup_loss, uaps_step_loss, uaps_pair_loss, unsup_scalars,
                     sup_scalars)
from .perturb import FeatureNoise, Dropout, FeatureDropout, manual_seed as perturb_manual_seed
from .metrics import mIoU, mDice, pixel_accuracy, seg_confusion, seg_confusion_per_image, metrics_from_confusion, mean_batch_metrics
from .unet import UNet, UNet_UAPS
from .res_uaps import ResUAPS, ResNet, resnet50
from .net_factory import net_factory
from .consistency import (softmax_mse_loss, softmax_kl_loss, kl_loss, entropy_map, entropy_minmization, uncertainty_map)
from .trainer import UAPSTrainer, BaselineTrainer
from . import augment, conv, data, dist, graph, inference, optim, sibling

__all__ = [n for n in dir() if not n.startswith("_")]

"""Loss helpers with the reference's names (utils.py:12-33 ``calculate_model_losses``, :139-146 ``add_loss``).

These run on whatever tensors they are given (torch ops on the GPU) and are the autograd-visible
spelling used by generic callers; the fused training iteration computes the same three terms
inside libsln_hip.so (csrc/vae_kernels.hip::loss_kernel).
"""
import torch
import torch.nn.functional as F


def add_loss(total_loss, curr_loss, loss_dict, loss_name, weight=1):
    weighted = curr_loss * weight
    loss_dict[loss_name] = weighted.item()
    return weighted if total_loss is None else total_loss + weighted


def calculate_model_losses(args, model, bbox, bbox_pred, angles, angles_pred, mu=None, logvar=None, KL_weight=None):
    losses = {}
    total = add_loss(0.0, F.l1_loss(bbox_pred, bbox), losses, 'bbox_pred', 1)
    total = add_loss(total, F.nll_loss(angles_pred, angles), losses, 'angle_pred', 1)
    if not args.use_AE:
        kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / mu.size(0)
        total = add_loss(total, kld, losses, 'KLD_Gauss', KL_weight)
    return total, losses


def get_model_attr(_object, attr):
    return getattr(_object.module if isinstance(_object, torch.nn.DataParallel) else _object, attr)

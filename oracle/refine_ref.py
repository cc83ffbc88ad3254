"""CPU restatement of the two torch-side pieces of the refinement loop that the product implements as HIP kernels.
TEST INFRASTRUCTURE ONLY (tests/ import it as the checker; the product never does).

  place_object / place_scene   models/diff_render.py:76-165 - box -> centre / size, theta = -angle * 2 pi / 24, isotropic
                               scale = min(size / model_size), R_y, translation, transformed vertices, size loss
  psp_pool / target_labels /   testing/test_render_refine.py:192-215 (PSP_pool_new), :328-356 (null fill, L1 * 0.5,
  refinement_loss              cross-entropy / 800 per scale, 100 * depth + 100 * semantic + 2 * size)

Parity unpinned: testing/test_render_refine.py and models/diff_render.py import neural_renderer / pymesh / pywavefront and the
SUNCG metadata at module level (models/misc.py:7-31) and cannot be imported in the build container, so no fixture can be
generated from them.  The functions below follow the cited lines statement by statement with the same torch calls
(F.interpolate for nn.Upsample / F.upsample, F.l1_loss, F.cross_entropy, F.mse_loss); meshes are passed in instead of being
retrieved from the SUNCG tables.
"""
import math

import torch
import torch.nn.functional as F

DO_NOT_VIS = ["wall", "ceiling", "floor", "person", "door", "window", "curtain", "blinds"]     # diff_render.py:93


def place_object(box, angle, room_ext, model_v, model_bbox_min, model_bbox_max):
    """diff_render.py:78-81,104,117-137: one object's vertices in room coordinates, and its size."""
    bbox_min, bbox_max = box[:3] * room_ext, box[3:] * room_ext                           # :78-79
    obj_center, obj_size = (bbox_max + bbox_min) / 2, bbox_max - bbox_min                 # :80-81
    theta = -angle * (2 * float(math.pi) / 24)                                            # :104
    model_size = model_bbox_max - model_bbox_min                                          # :110
    model_center = (model_bbox_min + model_bbox_max) / 2.0                                # :113
    scale = min([obj_size[0] / model_size[0], obj_size[1] / model_size[1], obj_size[2] / model_size[2]])     # :117
    rot = torch.eye(3, dtype=box.dtype)                                                   # :118-124
    cos_theta, sin_theta = torch.cos(theta), torch.sin(theta)
    rot = rot.clone()
    rot[0, 0] = cos_theta; rot[0, 2] = sin_theta; rot[2, 0] = -sin_theta; rot[2, 2] = cos_theta
    trans = obj_center - scale * torch.matmul(rot, model_center)                          # :126
    trans_4x4 = torch.eye(4, dtype=box.dtype); trans_4x4[:3, -1] = trans                  # :127-128
    rot_4x4 = torch.eye(4, dtype=box.dtype); rot_4x4[:3, :3] = rot * scale                # :129-130
    final = torch.matmul(trans_4x4, rot_4x4)[:3]                                          # :131
    v = torch.cat((model_v.t(), torch.ones(1, model_v.shape[0], dtype=box.dtype)), dim=0)            # :133-137
    return torch.t(torch.matmul(final, v)), obj_size


def place_scene(boxes, angles, class_names, models, obj_size_target=None):
    """The object loop of mesh_render_func (:76-159) over ``models`` = {class: dict(v, bbox_min, bbox_max)}: the concatenated
    object vertices (placement order), the sizes, and the size loss (:98-100).  The room row (last) is not placed."""
    verts, sizes = [], []
    size_loss = boxes.new_zeros(())
    k = 0
    for i, name in enumerate(class_names[:-1]):
        if name in DO_NOT_VIS or name not in models:
            continue
        m = models[name]
        v, size = place_object(boxes[i], angles[i], boxes[-1][3:], m["v"], m["bbox_min"], m["bbox_max"])
        if obj_size_target is not None:
            size_loss = size_loss + F.mse_loss(size, obj_size_target[k])
        verts.append(v); sizes.append(size)
        k += 1
    return (torch.cat(verts) if verts else boxes.new_zeros(0, 3)), sizes, size_loss


def psp_pool(feats, sizes=(32, 48, 64, 96), as_list=False):
    """PSP_pool_new (test_render_refine.py:192-215): nn.Upsample(size, 'bilinear', align_corners=True) per stage, then
    F.upsample(..., size=max, mode='bilinear') (align_corners defaults to False)."""
    outs = [F.interpolate(F.interpolate(feats, size=(s, s), mode='bilinear', align_corners=True), size=(sizes[-1], sizes[-1]),
                          mode='bilinear', align_corners=False) for s in sizes]
    return outs if as_list else torch.cat(outs, 1)


def target_labels(target, sizes=(32, 48, 64, 96)):
    """:336-343: per scale argmax of the pooled one-hot block, -100 where nothing is there."""
    out = []
    for pooled in psp_pool(target[:, 1:41], sizes, as_list=True):
        flat = torch.argmax(pooled, dim=1, keepdim=True)
        flat[torch.sum(pooled, dim=1, keepdim=True) < 0.5] = -100
        out.append(flat.detach())
    return out


def refinement_loss(iter_image, target, target_container, size_loss, sizes=(32, 48, 64, 96)):
    """:328-356.  Returns (loss_val, depth_loss, semantic_loss)."""
    iter_image = iter_image.clone()
    iter_image[:, -1][torch.sum(iter_image[:, 41:], dim=1) < 0.5] = 1.0                   # :329 fill in null regions
    scaled_target_depth = psp_pool(target[:, 41:], sizes)                                 # :331
    scaled_input_depth = psp_pool(iter_image[:, 41:], sizes)                              # :332
    train_labels_pooled = psp_pool(iter_image[:, 1:41], sizes, as_list=True)              # :335
    semantic_loss = iter_image.new_zeros(())
    for scale_idx in range(len(train_labels_pooled)):                                     # :346-347
        semantic_loss = semantic_loss + F.cross_entropy(train_labels_pooled[scale_idx], target_container[scale_idx][:, 0, :, :].long()) / 800.0
    depth_loss = F.l1_loss(scaled_input_depth, scaled_target_depth) * 0.5                 # :348 (orig_scaler 0.5)
    loss_val = depth_loss * 100 + semantic_loss * 100 + size_loss * 2.0                   # :350-352
    return loss_val, depth_loss, semantic_loss

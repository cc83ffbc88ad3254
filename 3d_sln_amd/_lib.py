"""ctypes binding of libsln_hip.so (the C ABI declared in include/sln_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
fails, an exception is raised.  PyTorch is only used by the callers for device
memory and streams; nothing here touches torch.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsln_hip.so")

SLN_E = {-1: "SLN_E_BADARG", -2: "SLN_E_UNSUPPORTED", -3: "SLN_E_STATE", -4: "SLN_E_NOGPU", -5: "SLN_E_NOMEM", -6: "SLN_E_CAPTURE"}

c_f32p = C.c_void_p
c_i64p = C.c_void_p


class SlnError(RuntimeError):
    pass


class SlnVaeConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "embedding_dim", "gconv_num_layers", "recurrent", "batch_norm", "decoder_cat", "use_ae",
        "box_dim", "n_angle", "num_objs", "num_preds", "num_attrs", "no_attr")]


class SlnVaeUnit(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "weight", "bias", "bn_weight", "bn_bias", "bn_running_mean", "bn_running_var",
        "bn_num_batches_tracked", "d_weight", "d_bias", "d_bn_weight", "d_bn_bias")]


EMB_NAMES = ("obj_emb_ec", "pred_emb_ec", "obj_emb_dc", "pred_emb_dc", "attr_emb_ec", "attr_emb_dc",
             "box_emb_w", "box_emb_b", "angle_emb")


class SlnVaeTensors(C.Structure):
    _fields_ = ([(p + n, C.c_void_p) for n in EMB_NAMES for p in ("", "d_")] +
                [("units_host", C.POINTER(SlnVaeUnit)),
                 ("flat_params", C.c_void_p), ("flat_grads", C.c_void_p),
                 ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("n_flat", C.c_int64)])


class SlnVaeBatch(C.Structure):
    _fields_ = [("objs", C.c_void_p), ("triples", C.c_void_p), ("boxes", C.c_void_p),
                ("angles", C.c_void_p), ("attributes", C.c_void_p), ("O", C.c_int), ("T", C.c_int)]


class SlnRoomTable(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("room_off", "cls", "bbox", "rot", "room_bbox", "room_id", "size_thr", "has_size")] + \
               [(n, C.c_int) for n in ("n_rooms", "n_classes", "use_attr_30", "reserved")]


class SlnGraphDraws(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("other", "swap", "attr_mode")]


class SlnGraphBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ids", "objs", "boxes", "triples", "angles", "attributes", "obj_to_img", "triple_to_img")]


class SlnPlacement(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n", "n_vis", "Vm", "Vs", "F", "reserved")] + \
               [(n, C.c_void_p) for n in ("vis", "model_v", "msize", "mcenter", "shell_v", "faces", "obj_face_ptr")] + \
               [("ext", C.c_float * 3), ("K", C.c_float * 9), ("R", C.c_float * 9), ("t", C.c_float * 3),
                ("orig_size", C.c_float), ("proj_eps", C.c_float), ("cull_eps", C.c_float)]


class SlnRefineLoss(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "image_size", "pooled_size", "channels", "sem0", "n_sem", "dep0", "n_dep", "n_scales",
                                       "stage1_stride")] + \
               [(n, C.c_void_p) for n in ("s2_k0", "s2_k1", "s2_l1", "s1_i0", "s1_i1", "s1_l1", "col_ptr", "col_out", "col_w")] + \
               [("max_col_entries", C.c_int), ("per_room", C.c_int), ("live_planes", C.c_void_p), ("null_mask", C.c_void_p), ("pooled_ones", C.c_void_p)]


class SlnPlacementRoom(C.Structure):
    _fields_ = [("P", SlnPlacement)] + \
               [(n, C.c_void_p) for n in ("boxes", "angles", "size_target", "faces_out", "sizes", "size_loss", "grad_faces", "grad_size_loss",
                                          "grad_boxes", "grad_angles", "boxes_pred", "angles_pred", "noise", "noise_step")] + \
               [("noise_stride", C.c_int64)] + [(n, C.c_void_p) for n in ("box_last", "angle_last", "grad_boxes_pred", "grad_angles_pred")] + \
               [("n_angle", C.c_int), ("ld_gb", C.c_int), ("beta", C.c_float), ("pad_", C.c_int)]


class SlnVaeGroupIO(C.Structure):
    _fields_ = [("rows_total", C.c_int), ("row0_host", C.POINTER(C.c_int))] + \
               [(n, C.c_void_p) for n in ("z", "boxes_pred", "angles_pred", "d_boxes_pred", "d_angles_pred", "dz", "sgd_step")]


# name -> (restype, argtypes); every symbol include/sln_hip.h declares must be listed here
# (tests/test_abi.py checks the header against this table and against the built library).
SIGNATURES = {
    "sln_version": (C.c_int, []),
    "sln_build_arch": (C.c_char_p, []),
    "sln_device_ok": (C.c_int, []),
    "sln_vae_num_units": (C.c_int, [C.POINTER(SlnVaeConfig)]),
    "sln_vae_create": (C.c_int, [C.POINTER(SlnVaeConfig), C.POINTER(C.c_void_p)]),
    "sln_vae_destroy": (None, [C.c_void_p]),
    "sln_vae_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "sln_vae_bind": (C.c_int, [C.c_void_p, C.POINTER(SlnVaeTensors), C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "sln_vae_set_batch": (C.c_int, [C.c_void_p, C.POINTER(SlnVaeBatch), C.c_void_p]),
    "sln_vae_check_batch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sln_vae_encoder": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_void_p]),
    "sln_vae_decoder": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p]),
    "sln_vae_forward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p]),
    "sln_vae_loss": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p, C.c_int, C.c_void_p]),
    "sln_vae_decoder_backward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sln_vae_encoder_backward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_void_p]),
    "sln_vae_backward": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sln_vae_params_changed": (C.c_int, [C.c_void_p]),
    "sln_vae_zero_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sln_vae_adam_step": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "sln_vae_adam_reset": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "sln_vae_adam_get_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "sln_vae_set_grad_guard": (C.c_int, [C.c_void_p, c_f32p]),
    "sln_vae_seed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sln_vae_last_eps": (C.c_int, [C.c_void_p, c_f32p, C.c_void_p]),
    "sln_vae_randn": (C.c_int, [C.c_void_p, c_f32p, C.c_int64, C.c_void_p]),
    "sln_layout_heatmap": (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "sln_vae_set_training": (C.c_int, [C.c_void_p, C.c_int]),
    "sln_vae_train_step": (C.c_int, [C.c_void_p, c_f32p, C.c_float, C.c_float, c_f32p, C.c_int, C.c_int, C.c_void_p]),
    "sln_vae_group_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(SlnVaeGroupIO), C.POINTER(C.c_void_p)]),
    "sln_vae_group_decoder": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sln_vae_group_decoder_backward": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sln_vae_group_launches": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sln_vae_group_transposes": (C.c_int64, [C.c_void_p]),
    "sln_vae_group_fused_params": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int]),
    "sln_vae_group_destroy": (None, [C.c_void_p]),
    "sln_debug_side_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sln_side_stream_prepare": (C.c_int, [C.c_void_p]),
    "sln_side_stream_forget": (C.c_int, [C.c_void_p]),
    "sln_prof_enable": (C.c_int, [C.c_int]),
    "sln_set_deterministic": (C.c_int, [C.c_int]),
    "sln_get_deterministic": (C.c_int, []),
    "sln_debug_tn_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "sln_prof_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "sln_vae_tap": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "sln_gconv_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sln_gconv_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SlnVaeUnit), c_f32p, c_f32p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, c_f32p, c_f32p, C.c_void_p]),
    "sln_gconv_net_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sln_gconv_net_set_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sln_gconv_net_forward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p]),
    "sln_gconv_net_backward": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sln_linear_forward": (C.c_int, [c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "sln_linear_wgrad": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_void_p]),
    "sln_project_faces": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_f32p, C.c_void_p]),
    "sln_project_faces_backward": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_f32p,
                                             c_f32p, C.c_void_p]),
    "sln_raster_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "sln_raster_forward": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                     c_f32p, c_f32p, C.c_void_p]),
    "sln_raster_forward_dual": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                          C.c_void_p, c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, C.c_void_p]),
    "sln_raster_texture_sample": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_float, c_f32p, C.c_void_p]),
    "sln_raster_backward_depth": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p,
                                            C.c_void_p]),
    "sln_raster_backward_rgb": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                          c_f32p, C.c_void_p]),
    "sln_raster_texture_sample_chw": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_float, c_f32p, C.c_void_p]),
    "sln_raster_backward_rgb_multi": (C.c_int, [c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                                C.c_void_p, c_f32p, C.c_void_p]),
    "sln_scene_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "sln_scene_forward": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, c_f32p, C.c_void_p]),
    "sln_scene_forward_live": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sln_spade_prepare": (C.c_int, [C.c_void_p]),
    "sln_spade_release": (C.c_int, [C.c_void_p, C.c_int]),
    "sln_spade_conv": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, c_f32p, C.c_void_p]),
    "sln_spade_modulate": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p,
                                     C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "sln_layernorm_stats": (C.c_int, [c_f32p, C.c_int, C.c_int64, C.c_float, C.c_void_p, c_f32p, C.c_void_p]),
    "sln_spade_apply": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "sln_spade_apply_up": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_float, c_f32p,
                                     C.c_void_p]),
    "sln_resize": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "sln_spade_depth_concat": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, C.c_void_p]),
    "sln_se_scale_add": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sln_spade_conv_sums": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_float, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sln_spade_modulate_up": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, C.c_int,
                                        c_f32p, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "sln_layernorm_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "sln_block_tail": (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_f32p, c_f32p, c_f32p,
                                 C.c_int, c_f32p, C.c_void_p, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "sln_upsample2x": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "sln_conv_img_tanh": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, C.c_void_p]),
    "sln_graph_plan": (C.c_int, [C.POINTER(SlnRoomTable), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sln_graph_draw": (C.c_int, [C.POINTER(SlnRoomTable), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "sln_graph_emit": (C.c_int, [C.POINTER(SlnRoomTable), C.c_void_p, C.c_int, C.c_void_p, C.POINTER(SlnGraphDraws),
                                 C.POINTER(SlnGraphBatch), C.c_void_p]),
    "sln_scene_live_channels": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sln_scene_backward": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_void_p, c_f32p, c_f32p, C.c_void_p]),
    "sln_place_forward": (C.c_int, [C.POINTER(SlnPlacement), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sln_place_backward": (C.c_int, [C.POINTER(SlnPlacement), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sln_refine_head_forward": (C.c_int, [C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    "sln_refine_head_backward": (C.c_int, [C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    "sln_refine_sgd": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_float, c_f32p, c_f32p, C.c_int64, C.c_float, C.c_void_p]),
    "sln_refine_head_forward_rooms": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p,
                                                c_f32p, C.c_void_p]),
    "sln_refine_head_backward_rooms": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p, C.c_int, c_f32p,
                                                 C.c_void_p]),
    "sln_place_forward_rooms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sln_place_backward_rooms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sln_refine_sgd_rooms": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.c_float, c_f32p, c_f32p,
                                       C.c_int64, C.c_float, C.c_void_p]),
    "sln_refine_loss_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sln_refine_loss_live_ok": (C.c_int, [C.POINTER(SlnRefineLoss)]),
    "sln_refine_loss_init": (C.c_int, [C.POINTER(SlnRefineLoss), C.c_void_p, C.c_void_p]),
    "sln_refine_pool": (C.c_int, [C.POINTER(SlnRefineLoss), c_f32p, C.c_int, C.c_void_p, c_f32p, C.c_void_p]),
    "sln_refine_loss_forward": (C.c_int, [C.POINTER(SlnRefineLoss), c_f32p, c_f32p, C.c_void_p, c_f32p, C.c_void_p, c_f32p, C.c_void_p]),
    "sln_refine_loss_backward": (C.c_int, [C.POINTER(SlnRefineLoss), C.c_void_p, c_f32p, c_f32p, C.c_void_p]),
}

_lib = None


def _load_hip_runtime():
    """libsln_hip.so is linked without a HIP runtime (build.py: -no-hip-rt) so that it binds to
    the one already in the process.  Under PyTorch-ROCm that is torch/lib/libamdhip64.so: promote
    it to the global symbol scope.  Without torch fall back to the system ROCm runtime."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:          # pragma: no cover - torch is always present in this image
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so"]
    for c in cands:
        if os.path.exists(c):
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return c
    raise SlnError("no HIP runtime (libamdhip64.so) found")


def lib():
    """Load the shared library (once).  Raises SlnError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlnError("libsln_hip.so is not built (%s); run `python __graft_entry__.py` or "
                           "`python 3d_sln_amd/build.py`. There is no CPU fallback." % LIB_PATH)
        _load_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError -> missing export: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    """Turn a C-ABI return code into an exception."""
    if rc == 0:
        return
    if rc < 0:
        raise SlnError("%s failed: %s" % (what, SLN_E.get(int(rc), "error %d" % rc)))
    raise SlnError("%s failed: hipError_t %d" % (what, rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

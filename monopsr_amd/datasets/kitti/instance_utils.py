"""Instance-map geometry, mirroring the TF functions of the reference's datasets/kitti/instance_utils.py that sit on
the model's output path (same names and argument meaning).  Tensors are torch CUDA tensors; the compute is
libmonopsr_hip.so (geometry.hip), wrapped in autograd Functions so the training loss can differentiate through it.
"""
import torch

from monopsr_amd import _lib


def _f32(t):
    return t.contiguous().float()


class _XyzLocalToGlobal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_local, view_angs, centroids):
        n = xyz_local.shape[0]
        p = xyz_local[0].numel() // 3 if n else 0
        xyz_local, view_angs, centroids = _f32(xyz_local), _f32(view_angs).reshape(-1), _f32(centroids)
        out = torch.empty_like(xyz_local)
        _lib.check(_lib.lib().mpsr_xyz_map_local_to_global(_lib.ptr(xyz_local), _lib.ptr(view_angs),
                                                           _lib.ptr(centroids), _lib.ptr(out), n, p, _lib.stream()))
        ctx.save_for_backward(view_angs)
        ctx.dims = (n, p)
        return out

    @staticmethod
    def backward(ctx, g):
        (view_angs,) = ctx.saved_tensors
        n, p = ctx.dims
        g = _f32(g)
        need_local, _, need_cen = ctx.needs_input_grad
        gl = torch.empty_like(g) if need_local else None
        gc = torch.empty((n, 3), dtype=torch.float32, device=g.device) if need_cen else None
        if need_local or need_cen:
            _lib.check(_lib.lib().mpsr_xyz_map_local_to_global_grad(_lib.ptr(g), _lib.ptr(view_angs), _lib.ptr(gl),
                                                                    _lib.ptr(gc), n, p, _lib.stream()))
        return gl, None, gc  # the viewing angle is an input of the graph (view_ang: 'est'), never a variable


def tf_inst_xyz_map_local_to_global(inst_xyz_map_local, map_roi_size, view_angs, centroids):
    """instance_utils.py:567-602.  (N,H,W,3) local map, (N,1) viewing angles, (N,3) centroids -> (N,H,W,3):
    every point rotated about y by the viewing angle, then translated by the centroid."""
    if inst_xyz_map_local.dim() != 4 or inst_xyz_map_local.shape[3] != 3:
        raise _lib.InvalidArgumentError("inst_xyz_map_local must be (N, H, W, 3)")
    n = inst_xyz_map_local.shape[0]
    if tuple(inst_xyz_map_local.shape[1:3]) != tuple(map_roi_size):
        raise _lib.InvalidArgumentError("inst_xyz_map_local does not match map_roi_size %s" % (tuple(map_roi_size),))
    if view_angs.numel() != n or tuple(centroids.shape) != (n, 3):
        raise _lib.InvalidArgumentError("view_angs must be (N, 1) and centroids (N, 3)")
    return _XyzLocalToGlobal.apply(inst_xyz_map_local, view_angs, centroids)


class _DepthLocalToGlobal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_local, global_depth, box_2d, inst_view_ang, cam_p, rotate_view):
        n, h, w = depth_local.shape[:3]
        # a (N,H,W,1) slice of an xyz map is read in place through its channel stride
        if depth_local.dim() == 4 and depth_local.stride(3) == 1 and depth_local.stride(2) > 1 and \
                depth_local.stride(1) == w * depth_local.stride(2) and depth_local.stride(0) == h * depth_local.stride(1) \
                and depth_local.dtype == torch.float32:
            src, stride = depth_local, depth_local.stride(2)
        else:
            src, stride = _f32(depth_local), 1
        z = _f32(global_depth).reshape(-1)
        out = torch.empty((n, h, w, 1), dtype=torch.float32, device=depth_local.device)
        boxes = _f32(box_2d) if rotate_view else None
        va = _f32(inst_view_ang).reshape(-1) if rotate_view else None
        cam = _f32(cam_p).reshape(-1) if rotate_view else None
        _lib.check(_lib.lib().mpsr_depth_map_local_to_global(src.data_ptr(), stride, _lib.ptr(z), _lib.ptr(boxes),
                                                             _lib.ptr(va), _lib.ptr(cam), _lib.ptr(out), n, h, w,
                                                             int(bool(rotate_view)), _lib.stream()))
        ctx.save_for_backward(*(t for t in (boxes, va, cam) if t is not None))
        ctx.meta = (n, h, w, bool(rotate_view), tuple(global_depth.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, rotate, zshape = ctx.meta
        boxes, va, cam = ctx.saved_tensors if rotate else (None, None, None)
        g = _f32(g)
        gz = None
        if ctx.needs_input_grad[1]:
            gz = torch.empty((n,), dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().mpsr_depth_map_local_to_global_grad(_lib.ptr(g), _lib.ptr(boxes), _lib.ptr(va),
                                                                      _lib.ptr(cam), _lib.ptr(gz), n, h, w,
                                                                      int(rotate), _lib.stream()))
            gz = gz.reshape(zshape)
        return (g if ctx.needs_input_grad[0] else None), gz, None, None, None, None


def tf_inst_depth_map_local_to_global(inst_depth_map_local, global_depth, box_2d=None, inst_view_ang=None,
                                      map_roi_size=None, cam_p=None, rotate_view=False):
    """instance_utils.py:605-680.  (N,H,W,1) local depth + (N,1) centroid depth -> (N,H,W,1) global depth; with
    rotate_view the view-normalisation offset is added (interpolated between the box's edge rays with H samples
    and laid out along the ROW axis, exactly as the reference does)."""
    if inst_depth_map_local.dim() != 4 or inst_depth_map_local.shape[3] != 1:
        raise _lib.InvalidArgumentError("inst_depth_map_local must be (N, H, W, 1)")
    if rotate_view and (box_2d is None or inst_view_ang is None or cam_p is None):
        raise _lib.InvalidArgumentError("rotate_view needs box_2d, inst_view_ang and cam_p")
    if map_roi_size is not None and tuple(inst_depth_map_local.shape[1:3]) != tuple(map_roi_size):
        raise _lib.InvalidArgumentError("inst_depth_map_local does not match map_roi_size")
    return _DepthLocalToGlobal.apply(inst_depth_map_local, global_depth, box_2d, inst_view_ang, cam_p, rotate_view)


class _ProjErrNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_global, boxes_2d, cam_p, valid_mask, want_maps):
        n, h, w = xyz_global.shape[:3]
        xyz_global, boxes_2d = _f32(xyz_global), _f32(boxes_2d)
        cam_p, valid_mask = _f32(cam_p).reshape(-1), _f32(valid_mask).reshape(n, h, w)
        norm = torch.empty((n,), dtype=torch.float32, device=xyz_global.device)
        maps = torch.empty((n, h, w, 2), dtype=torch.float32, device=xyz_global.device) if want_maps else None
        _lib.check(_lib.lib().mpsr_proj_err_norm(_lib.ptr(xyz_global), _lib.ptr(boxes_2d), _lib.ptr(cam_p),
                                                 _lib.ptr(valid_mask), _lib.ptr(maps), _lib.ptr(norm), n, h, w,
                                                 _lib.stream()))
        ctx.save_for_backward(xyz_global, boxes_2d, cam_p, valid_mask)
        if maps is None:
            maps = norm.new_empty(0)
        ctx.mark_non_differentiable(maps)
        return norm, maps

    @staticmethod
    def backward(ctx, gnorm, _gmaps):
        xyz_global, boxes_2d, cam_p, valid_mask = ctx.saved_tensors
        n, h, w = xyz_global.shape[:3]
        gx = torch.empty_like(xyz_global)
        _lib.check(_lib.lib().mpsr_proj_err_norm_grad(_lib.ptr(_f32(gnorm)), _lib.ptr(xyz_global), _lib.ptr(boxes_2d),
                                                      _lib.ptr(cam_p), _lib.ptr(valid_mask), _lib.ptr(gx), n, h, w,
                                                      _lib.stream()))
        return gx, None, None, None, None


def proj_err_maps_norm(pred_inst_xyz_map_global, pred_boxes_2d, cam_p, valid_mask_maps, want_maps=False):
    """The arithmetic of monopsr_output_builder.py:681-746 (get_proj_err_maps_norm) -> (proj_err_norm (N,),
    proj_err_maps_norm (N,H,W,2) or None).  Differentiable w.r.t. the global map."""
    if pred_inst_xyz_map_global.dim() != 4 or pred_inst_xyz_map_global.shape[3] != 3:
        raise _lib.InvalidArgumentError("pred_inst_xyz_map_global must be (N, H, W, 3)")
    n = pred_inst_xyz_map_global.shape[0]
    if tuple(pred_boxes_2d.shape) != (n, 4) or cam_p.numel() != 12 or \
            valid_mask_maps.numel() != pred_inst_xyz_map_global.numel() // 3:
        raise _lib.InvalidArgumentError("boxes_2d must be (N,4), cam_p (3,4), valid_mask_maps (N,H,W,1)")
    norm, maps = _ProjErrNorm.apply(pred_inst_xyz_map_global, pred_boxes_2d, cam_p, valid_mask_maps, want_maps)
    return norm, (maps if want_maps else None)


def format_boxes(lwh, view_angs, alpha_bins, alpha_regs, centroids, boxes_2d, scores, class_indices, cam_p,
                 img_shape, centroid_type='middle', post_process_cen_x=True, max_depth=45.0):
    """monopsr_model.py:960-1071 format_predictions' box arithmetic, incl. postprocess_cen_x
    (instance_utils.py:988-1032) and score_boxes (monopsr_output_builder.py:805-860), on the device in fp64:
    -> (box_3d (N,9) [x,y,z,l,w,h,ry,score,class-1], box_2d (N,7) [y1,x1,y2,x2,alpha,score,class-1])."""
    n, nb = alpha_bins.shape
    dev = alpha_bins.device
    b3 = torch.empty((n, 9), dtype=torch.float32, device=dev)
    b2 = torch.empty((n, 7), dtype=torch.float32, device=dev)
    args = [_f32(t) for t in (lwh, view_angs.reshape(-1), alpha_bins, alpha_regs, centroids, boxes_2d,
                              scores.reshape(-1))]
    cls = class_indices.reshape(-1).contiguous().int()
    cam = _f32(cam_p).reshape(-1)
    _lib.check(_lib.lib().mpsr_format_boxes(*[_lib.ptr(t) for t in args], _lib.ptr(cls), _lib.ptr(cam), n, nb,
                                            int(img_shape[0]), int(img_shape[1]), int(centroid_type == 'middle'),
                                            int(bool(post_process_cen_x)), float(max_depth), _lib.ptr(b3),
                                            _lib.ptr(b2), _lib.stream()))
    return b3, b2

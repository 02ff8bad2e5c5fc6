"""torch-facing wrappers over the C ABI: device tensors in, device tensors out.

Everything here runs the hand-written HIP kernels of libtuch_amd.so; there is no
eager/CPU fallback.  Index outputs are int64 like the reference's torch ops.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.autograd.function import once_differentiable

from . import _C

MODE_SMPLIFY = 0   # tuch/smplify/losses.py:96-105
MODE_TRAIN = 1     # tuch/train/loss.py:303-315


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype is torch.float32 and t.is_contiguous():       # (the usual case: one call instead of three on the eager path)
        return t.detach()
    return t.detach().to(torch.float32).contiguous()


_SIDE_STREAMS = {}


_STREAM_OBJECTS = {}


def _current_stream(device) -> 'torch.cuda.Stream':
    """torch.cuda.current_stream(device) without building a new Stream object through torch's device-index helpers on
    every call (~9 us, several times per eager step): one object per (device, raw stream handle) is kept."""
    raw, cur = _C._raw_stream, _C._cur_device
    if raw is None or cur is None:
        return torch.cuda.current_stream(device)
    index = device.index if device.index is not None else cur()
    key = (index, raw(index))
    s = _STREAM_OBJECTS.get(key)
    if s is None:
        s = _STREAM_OBJECTS[key] = torch.cuda.current_stream(device)
    return s


def _side_stream(device) -> 'torch.cuda.Stream':
    """The companion stream of the CURRENT stream (one per stream: callers that run several fits on streams of
    their own must not be coupled through a shared side stream)."""
    key = (device.type, device.index, _current_stream(device).cuda_stream)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


_GRAPH_STREAMS = {}


@contextlib.contextmanager
def off_default_stream(device):
    """Run the body on a non-default stream when the current stream is the legacy default (NULL) stream.

    hipGraph replays on the NULL stream are not reliably ordered against the kernels, copies and memsets enqueued
    around them (ROCm 7.2, graphs with forked branches: observed as stale inputs / un-reset optimiser state in about
    every second call, never on a created stream).  Every place in this package that replays a captured graph does it
    inside this context; the two streams are joined with events on entry and exit, so callers see stream semantics."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    if cur != torch.cuda.default_stream(device):
        yield cur
        return
    key = (device.index if device.index is not None else torch.cuda.current_device())
    s = _GRAPH_STREAMS.get(key)
    if s is None:
        s = _GRAPH_STREAMS[key] = torch.cuda.Stream(device=device)
    s.wait_stream(cur)
    try:
        with torch.cuda.stream(s):
            yield s
    finally:
        cur.wait_stream(s)


def deterministic() -> bool:
    """Deterministic mode of the library (include/tuch_amd.h: tuch_set_deterministic): on unless TUCH_DETERMINISTIC=0."""
    return bool(_C.lib().tuch_get_deterministic())


@contextlib.contextmanager
def deterministic_mode(on: bool):
    """``with ops.deterministic_mode(False): ...`` -- the mode for the calls inside, the previous one restored after."""
    before = deterministic()
    set_deterministic(on)
    try:
        yield
    finally:
        set_deterministic(before)


def set_deterministic(on: bool) -> None:
    """True (the default): gradient scatters through 64-bit fixed-point integer atomics instead of float atomics: an SMPLify-DC fit reproduces
    bit for bit (stage-2 tail, SMPL adjoint, the training loss's plain term -- _ContactTerms -- and its HD term).
    Graphs captured before the switch keep the mode they were captured in."""
    _C.lib().tuch_set_deterministic(int(bool(on)))


_ONES = {}
_ROOT_NODES = {}     # id(node) -> node: the autograd nodes of the losses of the RUNNING backward_scalar() calls (a plain dict:
                     # the engine runs a device's nodes on its own worker thread, a thread-local would be invisible there)


def backward_scalar(loss: torch.Tensor) -> None:
    """``loss.backward()`` for a scalar loss with the seed gradient (ones) taken from a cache: autograd otherwise fills a
    fresh one per call, a launch of its own at the head of every backward chain.  The loss's own autograd node is
    remembered for the duration of the pass: a node that finds ITSELF there is the root of the pass -- every gradient of
    the pass flows through what it returns (what ops._Stage2Tail needs to know before it lets the body model's backward
    kernel apply the optimiser's update; the seed's address alone does not say so: AddBackward hands the same tensor on)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    seed = _ONES.get(key)
    if seed is None:
        seed = _ONES[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    node = loss.grad_fn
    if node is not None:
        _ROOT_NODES[id(node)] = node
    try:
        loss.backward(gradient=seed)
    finally:
        if node is not None:
            _ROOT_NODES.pop(id(node), None)


def is_root_of_backward_scalar(node) -> bool:
    """True inside a backward pass started by backward_scalar() on the output of exactly this autograd node."""
    return node is not None and _ROOT_NODES.get(id(node)) is node


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------- raw functions
class _PairwiseDist(torch.autograd.Function):
    """tuch/utils/contact.py:23-47 with its gradient (the reference differentiates through the matrix:
    smplify/losses.py:76-78 -> 115-116, eft/loss.py:142).  One kernel forward, one per differentiated input backward."""

    @staticmethod
    def forward(ctx, x, y, squared):
        xf, yf = _f32(x), _f32(y)
        b, nx, _ = xf.shape
        ny = yf.shape[1]
        out = torch.empty(b, nx, ny, dtype=torch.float32, device=xf.device)
        _C.check(_C.lib().tuch_batch_pairwise_dist(_C.ptr(xf), _C.ptr(yf), b, nx, ny, int(squared), _C.ptr(out),
                                                   _C.stream()))
        ctx.save_for_backward(xf, yf)
        ctx.squared = bool(squared)
        ctx.dtypes = (x.dtype, y.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        xf, yf = ctx.saved_tensors
        b, nx, _ = xf.shape
        ny = yf.shape[1]
        g = grad_out.to(torch.float32).contiguous()
        gx = torch.empty_like(xf) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(yf) if ctx.needs_input_grad[1] else None
        if gx is not None or gy is not None:
            _C.check(_C.lib().tuch_batch_pairwise_dist_bwd(_C.ptr(xf), _C.ptr(yf), _C.ptr(g), b, nx, ny, int(ctx.squared),
                                                           _C.ptr(gx), _C.ptr(gy), _C.stream()))
        return (None if gx is None else gx.to(ctx.dtypes[0]), None if gy is None else gy.to(ctx.dtypes[1]), None)


def batch_pairwise_dist(x: torch.Tensor, y: torch.Tensor, squared: bool = True) -> torch.Tensor:
    return _PairwiseDist.apply(x, y, squared)


def _solid_angles_raw(points: torch.Tensor, triangles: torch.Tensor) -> torch.Tensor:
    points, triangles = _f32(points), _f32(triangles)
    b, q, _ = points.shape
    f = triangles.shape[1]
    out = torch.empty(b, q, f, dtype=torch.float32, device=points.device)
    _C.check(_C.lib().tuch_solid_angles(_C.ptr(points), _C.ptr(triangles), b, q, f, _C.ptr(out), _C.stream()))
    return out


def _winding_numbers_raw(points: torch.Tensor, triangles: torch.Tensor, thresh: Optional[float] = None):
    points, triangles = _f32(points), _f32(triangles)
    b, q, _ = points.shape
    f = triangles.shape[1]
    L = _C.lib()
    w = torch.empty(b, q, dtype=torch.float32, device=points.device)
    ext = torch.empty(b, q, dtype=torch.uint8, device=points.device) if thresh is not None else None
    nbytes = L.tuch_winding_workspace_bytes(b, q, f)
    ws = _workspace(nbytes, points.device)
    _C.check(L.tuch_winding_numbers(_C.ptr(points), _C.ptr(triangles), b, q, f, _C.ptr(w), _C.ptr(ext),
                                    float(thresh if thresh is not None else 0.0), _C.ptr(ws), nbytes,
                                    _C.stream()))
    return (w, ext.bool()) if thresh is not None else w


class _SolidAngles(torch.autograd.Function):
    """solid_angles / winding_numbers (tuch/utils/contact.py:49-147) with their gradient: plain differentiable torch ops in the
    reference (which itself only calls them under torch.no_grad()), two adjoint kernels here (csrc/solid_angle_bwd.hip)."""

    @staticmethod
    def forward(ctx, points, triangles, winding):
        pf, tf = _f32(points), _f32(triangles)
        ctx.save_for_backward(pf, tf)
        ctx.winding = bool(winding)
        ctx.dtypes = (points.dtype, triangles.dtype)
        return _winding_numbers_raw(pf, tf) if winding else _solid_angles_raw(pf, tf)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        pf, tf = ctx.saved_tensors
        b, q, _ = pf.shape
        f = tf.shape[1]
        g = grad_out.to(torch.float32).contiguous()
        gp = torch.empty_like(pf) if ctx.needs_input_grad[0] else None
        gt = torch.empty_like(tf) if ctx.needs_input_grad[1] else None
        if gp is not None or gt is not None:
            _C.check(_C.lib().tuch_solid_angles_bwd(_C.ptr(pf), _C.ptr(tf), None if ctx.winding else _C.ptr(g),
                                                    _C.ptr(g) if ctx.winding else None, b, q, f, _C.ptr(gp), _C.ptr(gt),
                                                    _C.stream()))
        return (None if gp is None else gp.to(ctx.dtypes[0]), None if gt is None else gt.to(ctx.dtypes[1]), None)


def _tracked(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


def solid_angles(points: torch.Tensor, triangles: torch.Tensor) -> torch.Tensor:
    if _tracked(points, triangles):
        return _SolidAngles.apply(points, triangles, False)
    return _solid_angles_raw(points, triangles)


def winding_numbers(points: torch.Tensor, triangles: torch.Tensor, thresh: Optional[float] = None):
    """[B,Q,3], [B,F,3,3] -> w [B,Q] (and exterior = w <= thresh if thresh is given)."""
    if thresh is None and _tracked(points, triangles):
        return _SolidAngles.apply(points, triangles, True)
    return _winding_numbers_raw(points, triangles, thresh)


def gather_triangles(verts: torch.Tensor, faces_i32: torch.Tensor) -> torch.Tensor:
    verts = _f32(verts)
    b, v, _ = verts.shape
    f = faces_i32.shape[0]
    out = torch.empty(b, f, 3, 3, dtype=torch.float32, device=verts.device)
    _C.check(_C.lib().tuch_gather_triangles(_C.ptr(verts), _C.ptr(faces_i32), b, v, f, _C.ptr(out),
                                            _C.stream()))
    return out


def pack_geomask(geomask: torch.Tensor) -> torch.Tensor:
    """[V,V] bool on device -> bit-packed words int64 [W,V] (layout of v2v.hip)."""
    gm = geomask.to(torch.uint8).contiguous()
    v = gm.shape[0]
    L = _C.lib()
    bits = torch.empty(L.tuch_geomask_words(v), v, dtype=torch.int64, device=gm.device)
    _C.check(L.tuch_pack_geomask(_C.ptr(gm), v, _C.ptr(bits), _C.stream()))
    return bits


def v2v_min_masked(points: torch.Tensor, mask_bits_ptr, num_points: Optional[int] = None):
    """Masked nearest neighbour: [B,N,3] + packed mask -> (min_d2 [B,N] f32, argmin [B,N] int32)."""
    points = _f32(points)
    b, n, _ = points.shape
    L = _C.lib()
    mn = torch.empty(b, n, dtype=torch.float32, device=points.device)
    arg = torch.empty(b, n, dtype=torch.int32, device=points.device)
    nbytes = L.tuch_v2v_workspace_bytes(b, n)
    ws = _workspace(nbytes, points.device)
    mptr = _C.ptr(mask_bits_ptr) if isinstance(mask_bits_ptr, torch.Tensor) else mask_bits_ptr
    _C.check(L.tuch_v2v_min_masked(_C.ptr(points), mptr, b, n, _C.ptr(mn), _C.ptr(arg), _C.ptr(ws), nbytes,
                                   _C.stream()))
    return mn, arg


class _ContactTerms(torch.autograd.Function):
    """terms[b] = (interior sum, exterior sum); gradient flows to the points only."""

    @staticmethod
    def forward(ctx, points, partner, exterior, valid, mode, euclthres, grad_masked=False):
        pts = _f32(points)
        b, n, _ = pts.shape
        terms = torch.empty(b, 2, dtype=torch.float32, device=pts.device)
        _C.check(_C.lib().tuch_contact_terms_fwd(_C.ptr(pts), _C.ptr(partner), _C.ptr(exterior), _C.ptr(valid),
                                                 b, n, int(mode), float(euclthres), _C.ptr(terms), _C.stream()))
        ctx.save_for_backward(pts, partner, exterior, valid)
        ctx.mode, ctx.euclthres = int(mode), float(euclthres)
        ctx.grad_masked = bool(grad_masked)     # the caller reduces the terms with valid_mean: its gradient is 0 for the others
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        pts, partner, exterior, valid = ctx.saved_tensors
        b, n, _ = pts.shape
        g = grad_terms.to(torch.float32)
        if valid is not None and not ctx.grad_masked:
            g = g * valid.to(g.dtype)[:, None]
        g = g.contiguous()   # [B,2]: upstream gradient of the interior and of the exterior sum
        if deterministic():
            # 64-bit fixed-point accumulators + integer atomics: the plain training term's gradient is bit-reproducible too
            fixed = torch.zeros(b * n * 3, dtype=torch.int64, device=pts.device)
            grad = torch.empty_like(pts)
            _C.check(_C.lib().tuch_contact_terms_bwd_fixed(_C.ptr(pts), _C.ptr(partner), _C.ptr(exterior), _C.ptr(g), b, n,
                                                           ctx.mode, ctx.euclthres, _C.ptr(fixed), _C.ptr(grad), _C.stream()))
            return grad, None, None, None, None, None, None
        grad = torch.zeros_like(pts)
        _C.check(_C.lib().tuch_contact_terms_bwd(_C.ptr(pts), _C.ptr(partner), _C.ptr(exterior), _C.ptr(g),
                                                 b, n, ctx.mode, ctx.euclthres, _C.ptr(grad), _C.stream()))
        return grad, None, None, None, None, None, None


class _HDPoints(torch.autograd.Function):
    """points[n] = sum_k w[hd[n],k] * verts[body[n], idx[hd[n],k]]  (loss.py:285 with the regressor's 3 non-zeros per row)."""

    @staticmethod
    def forward(ctx, verts, body, hd, idx, w):
        v = _f32(verts)
        n = body.shape[0]
        out = torch.empty(n, 3, dtype=torch.float32, device=v.device)
        _C.check(_C.lib().tuch_hd_points_fwd(_C.ptr(v), _C.ptr(body), _C.ptr(hd), _C.ptr(idx), _C.ptr(w), v.shape[1], n,
                                             _C.ptr(out), _C.stream()))
        ctx.save_for_backward(body, hd, idx, w)
        ctx.shape = v.shape
        return out

    @staticmethod
    def backward(ctx, grad):
        body, hd, idx, w = ctx.saved_tensors
        g = grad.to(torch.float32).contiguous()
        gv = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        _C.check(_C.lib().tuch_hd_points_bwd(_C.ptr(g), _C.ptr(body), _C.ptr(hd), _C.ptr(idx), _C.ptr(w), ctx.shape[1],
                                             body.shape[0], _C.ptr(gv), _C.stream()))
        return gv, None, None, None, None


def hd_points(verts, body_of_point, hd_of_point, hd_idx, hd_w):
    """Selected HD points [N,3] of posed vertices [B,V,3]; int32 indices, hd_idx / hd_w [N_hd,3]."""
    return _HDPoints.apply(verts, body_of_point, hd_of_point, hd_idx, hd_w)


class _ContactTermsRagged(torch.autograd.Function):
    """terms[b] over the ragged point set of body b (HD points): points [N,3], global partner indices."""

    @staticmethod
    def forward(ctx, points, partner, exterior, offsets, body_of_point, mode, euclthres):
        pts = _f32(points)
        b = offsets.shape[0] - 1
        terms = torch.empty(b, 2, dtype=torch.float32, device=pts.device)
        _C.check(_C.lib().tuch_contact_terms_ragged_fwd(_C.ptr(pts), _C.ptr(partner), _C.ptr(exterior),
                                                        _C.ptr(offsets), b, int(mode), float(euclthres),
                                                        _C.ptr(terms), _C.stream()))
        ctx.save_for_backward(pts, partner, exterior, body_of_point)
        ctx.mode, ctx.euclthres = int(mode), float(euclthres)
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        pts, partner, exterior, body_of_point = ctx.saved_tensors
        grad = torch.zeros_like(pts)
        g = grad_terms.to(torch.float32).contiguous()
        _C.check(_C.lib().tuch_contact_terms_ragged_bwd(_C.ptr(pts), _C.ptr(partner), _C.ptr(exterior),
                                                        _C.ptr(body_of_point), _C.ptr(g), pts.shape[0], ctx.mode,
                                                        ctx.euclthres, _C.ptr(grad), _C.stream()))
        return grad, None, None, None, None, None, None


def contact_terms_ragged(points, partner_i32, exterior_u8, offsets_i32, body_of_point_i32, mode, euclthres):
    """[B,2] (interior sum, exterior sum) per body over a concatenated ragged point set."""
    return _ContactTermsRagged.apply(points, partner_i32, exterior_u8, offsets_i32, body_of_point_i32, mode,
                                     euclthres)


def contact_terms(points, partner_i32, exterior_u8, valid_u8, mode, euclthres):
    """Sum of pull/push terms per body: [B] = interior + exterior (differentiable wrt points)."""
    terms = _ContactTerms.apply(points, partner_i32, exterior_u8, valid_u8, mode, euclthres)
    return terms.sum(dim=1), terms


class _ValidMean(torch.autograd.Function):
    """sum of terms [B,K] over the valid bodies / their number (loss.py:317), one launch each way."""

    @staticmethod
    def forward(ctx, terms, valid_u8):
        t = _f32(terms)
        b = t.shape[0]
        k = t.numel() // b
        out = torch.empty(2, dtype=torch.float32, device=t.device)
        _C.check(_C.lib().tuch_valid_mean_fwd(_C.ptr(t), _C.ptr(valid_u8), b, k, _C.ptr(out), _C.stream()))
        ctx.save_for_backward(out, valid_u8)
        ctx.shape = tuple(terms.shape)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        out, valid_u8 = ctx.saved_tensors
        b = ctx.shape[0]
        k = int(torch.Size(ctx.shape).numel()) // b
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=out.device)
        up = g.to(torch.float32).reshape(1).contiguous()
        _C.check(_C.lib().tuch_valid_mean_bwd(_C.ptr(up), _C.ptr(out), _C.ptr(valid_u8), b, k, _C.ptr(grad), _C.stream()))
        return grad, None


def as_u8(mask):
    """A [B] boolean / byte mask as uint8 without a copy where the bytes are already there."""
    if mask.dtype == torch.bool and mask.is_contiguous():
        return mask.view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


def valid_mean(terms, valid_u8):
    """Scalar: terms [B,...] summed over the bodies with valid != 0, divided by their number (NaN when there is none, like
    the reference's ``loss.sum() / valid_fit.sum()``, loss.py:317).  The gradient of the other bodies' terms is 0."""
    return _ValidMean.apply(terms, valid_u8)


def contact_terms_mean(points, partner_i32, exterior_u8, valid_u8, mode, euclthres):
    """``contact_terms(...)[0].sum() / valid.sum()`` of RegressorLoss.contact_loss (loss.py:259-272, 317) in two launches
    forward and two backward."""
    terms = _ContactTerms.apply(points, partner_i32, exterior_u8, valid_u8, mode, euclthres, True)
    return valid_mean(terms, valid_u8)


class _SmallTerms(torch.autograd.Function):
    """Reprojection + max-mixture prior of the SMPLify-DC objective in one kernel; returns [B,2]."""

    @staticmethod
    def forward(ctx, joints, camera_t, body_pose, camera_center, joints_2d, joints_conf, means, precisions,
                log_weights, focal, sigma, prior_scale):
        j = _f32(joints)
        b, nj, _ = j.shape
        out = torch.empty(b, 2, dtype=torch.float32, device=j.device)
        gj = torch.empty_like(j)
        gc = torch.empty(b, 3, dtype=torch.float32, device=j.device)
        gp = torch.empty(b, 69, dtype=torch.float32, device=j.device) if body_pose is not None else None
        # bound to locals: a converted copy must stay alive until the kernel has been enqueued
        cam_t, cam_c, j2d, conf = _f32(camera_t), _f32(camera_center), _f32(joints_2d), _f32(joints_conf)
        pose = _f32(body_pose) if body_pose is not None else None
        _C.check(_C.lib().tuch_smplify_small_terms(
            _C.ptr(j), _C.ptr(cam_t), _C.ptr(cam_c), _C.ptr(j2d), _C.ptr(conf), _C.ptr(pose),
            _C.ptr(means), _C.ptr(precisions), _C.ptr(log_weights), b, nj,
            means.shape[0] if means is not None else 0, float(focal), float(sigma), float(prior_scale),
            _C.ptr(out), _C.ptr(gj), _C.ptr(gc), _C.ptr(gp), _C.stream()))
        ctx.save_for_backward(gj, gc, gp)
        return out

    @staticmethod
    def backward(ctx, g):
        gj, gc, gp = ctx.saved_tensors
        g_rep, g_pri = g[:, 0].contiguous(), g[:, 1].contiguous()
        return (gj * g_rep[:, None, None], gc * g_rep[:, None],
                gp * g_pri[:, None] if gp is not None else None) + (None,) * 9


def smplify_small_terms(joints, camera_t, body_pose, camera_center, joints_2d, joints_conf, means, precisions,
                        log_weights, focal, sigma, prior_scale):
    return _SmallTerms.apply(joints, camera_t, body_pose, camera_center, joints_2d, joints_conf, means,
                             precisions, log_weights, focal, sigma, prior_scale)


class _Objective(torch.autograd.Function):
    """sum(small) + contact_scale * sum(terms) + r2r_scale * sum(r2r) as one deterministic kernel."""

    @staticmethod
    def forward(ctx, small, terms, r2r, contact_scale, r2r_scale):
        small, terms = small.contiguous(), terms.contiguous()
        b = small.shape[0]
        p = r2r.shape[1] if r2r is not None else 0
        out = torch.empty(1, dtype=torch.float32, device=small.device)
        r2r_c = r2r.contiguous() if p else None
        _C.check(_C.lib().tuch_smplify_objective(_C.ptr(small), _C.ptr(terms), _C.ptr(r2r_c), b, p,
                                                 float(contact_scale), float(r2r_scale), _C.ptr(out), _C.stream()))
        ctx.shape = (b, p, float(contact_scale), float(r2r_scale))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        b, p, cs, rs = ctx.shape
        g = g.reshape(1).to(torch.float32).contiguous()
        gs = torch.empty(b, 2, dtype=torch.float32, device=g.device)
        gt = torch.empty(b, 2, dtype=torch.float32, device=g.device)
        gr = torch.empty(b, p, dtype=torch.float32, device=g.device) if p else None
        _C.check(_C.lib().tuch_smplify_objective_bwd(_C.ptr(g), b, p, cs, rs, _C.ptr(gs), _C.ptr(gt), _C.ptr(gr),
                                                     _C.stream()))
        return gs, gt, gr, None, None


def smplify_objective(small, terms, r2r, contact_scale, r2r_scale):
    return _Objective.apply(small, terms, r2r, contact_scale, r2r_scale)


_TICKETS = {}


def _ticket(device) -> torch.Tensor:
    """A zeroed int: the arrival counter of kernels whose last block finishes the job (they leave it zero).
    Eager calls: one per (device, stream) -- calls on one stream are ordered, so they can share it; another stream gets its
    own.  Under hipGraph capture: a counter of the call's own, allocated from the capturing graph's pool and cleared by a
    memset node of that graph -- a captured graph may be replayed on ANY stream, beside other graphs that were captured on
    the same (shared) capture stream, so a counter keyed by the capture stream would be shared by launches that are not
    ordered against each other (and would live in the pool of whichever graph happened to be captured first)."""
    if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
        return torch.zeros(1, dtype=torch.int32, device=device)
    key = (device.index, _current_stream(device).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def _rows_of(t, b, tail, name):
    """float32, contiguous, [b, *tail]; a batch of one is broadcast like the torch-op form would (the kernels index raw
    pointers by body: any other shape would be an out-of-bounds read)."""
    t = _f32(t)
    if t.dim() == len(tail) + 1 and t.shape[0] == 1 and b > 1:
        t = t.expand(b, *t.shape[1:]).contiguous()
    if tuple(t.shape) != (b,) + tuple(tail):
        raise ValueError('stage-1 objective: %s has shape %s, expected %s' % (name, tuple(t.shape), (b,) + tuple(tail)))
    return t


class _Stage1Objective(torch.autograd.Function):
    """camera_fitting_loss (tuch/smplify/losses.py:125-152) as ONE launch: reprojection + depth term + shape prior, the
    scalar total and the unit gradients w.r.t. joints / camera translation / betas (csrc/small_terms.hip:
    stage1_terms_kernel).  backward() only hands the gradients over (scaled unless the upstream gradient is the cached
    unit seed of ops.backward_scalar)."""

    @staticmethod
    def forward(ctx, joints, camera_t, betas, camera_t_est, camera_center, joints_2d, joints_conf, focal, sigma, depth_w,
                shape_w):
        j = _f32(joints)
        if j.dim() != 3 or j.shape[2] != 3:
            raise ValueError('stage-1 objective: joints must be [B,J,3], got %s' % (tuple(j.shape),))
        b, nj, _ = j.shape
        ct = _rows_of(camera_t, b, (3,), 'camera_t')
        be = None
        if betas is not None and shape_w != 0.0:
            if betas.dim() != 2:
                raise ValueError('stage-1 objective: betas must be [B,n], got %s' % (tuple(betas.shape),))
            be = _rows_of(betas, b, (betas.shape[1],), 'betas')
        est, cc = _rows_of(camera_t_est, b, (3,), 'camera_t_est'), _rows_of(camera_center, b, (2,), 'camera_center')
        j2d, conf = _rows_of(joints_2d, b, (nj, 2), 'joints_2d'), _rows_of(joints_conf, b, (nj,), 'joints_conf')
        out = torch.empty(1, dtype=torch.float32, device=j.device)
        share = torch.empty(b, dtype=torch.float32, device=j.device)
        gj = torch.empty_like(j)
        gc = torch.empty(b, 3, dtype=torch.float32, device=j.device)
        gb = torch.empty_like(be) if be is not None else None
        _C.check(_C.lib().tuch_smplify_stage1_terms(
            _C.ptr(j), _C.ptr(ct), _C.ptr(est), _C.ptr(cc), _C.ptr(j2d), _C.ptr(conf), _C.ptr(be), b, nj,
            be.shape[1] if be is not None else 0, float(focal), float(sigma), float(depth_w), float(shape_w), _C.ptr(share),
            _C.ptr(_ticket(j.device)), _C.ptr(out), _C.ptr(gj), _C.ptr(gc), _C.ptr(gb), _C.stream()))
        ctx.save_for_backward(gj, gc, gb)
        ctx.in_dtypes = (joints.dtype, camera_t.dtype, betas.dtype if betas is not None else None)
        ctx.in_shapes = (tuple(camera_t.shape), tuple(betas.shape) if betas is not None else None)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        gj, gc, gb = ctx.saved_tensors
        if not any(g.data_ptr() == seed.data_ptr() for seed in _ONES.values()):
            g = g.reshape(()).to(torch.float32)
            gj, gc, gb = gj * g, gc * g, (gb * g if gb is not None else None)
        dj, dc, db = ctx.in_dtypes
        sc, sb = ctx.in_shapes
        if tuple(gc.shape) != sc:                     # a broadcast batch of one: its gradient is the sum over the bodies
            gc = gc.sum(0, keepdim=True)
        if gb is not None and tuple(gb.shape) != sb:
            gb = gb.sum(0, keepdim=True)
        return (gj.to(dj), gc.to(dc), gb.to(db) if gb is not None else None) + (None,) * 8


def smplify_stage1_objective(joints, camera_t, betas, camera_t_est, camera_center, joints_2d, joints_conf, focal, sigma,
                             depth_weight, shape_weight):
    return _Stage1Objective.apply(joints, camera_t, betas, camera_t_est, camera_center, joints_2d, joints_conf, focal, sigma,
                                  depth_weight, shape_weight)


def _graph_task_id() -> int:
    """id of the running autograd backward pass (-1 outside one, or where this torch build has no such query)"""
    f = getattr(torch._C, '_current_graph_task_id', None)
    return int(f()) if f is not None else -1


class _Stage2Tail(torch.autograd.Function):
    """Everything of the stage-2 objective behind the body model as ONE autograd node (tuch/smplify/losses.py:56-123):
    inside test + nearest admissible vertex (no gradient), contact sums, region minima, reprojection + prior, the
    weighted total.  The same kernels as the separate nodes (_SmallTerms, _ContactTerms, _RegionPairMin, _Objective);
    what goes away is the torch glue between them -- in the backward pass ~10 small launches per step (scalings by the
    upstream gradient, dtype / layout copies, a zero fill and an add per gradient path into the vertices).  Backward:
    one kernel turns the upstream scalar into the weights of every term, then the contact and region kernels
    accumulate into one vertex gradient."""

    @staticmethod
    def forward(ctx, verts, joints, camera_t, body_pose, model, valid, select, const):
        v, j = _f32(verts), _f32(joints)
        cam_t, pose = _f32(camera_t), _f32(body_pose)
        b, nj, _ = j.shape
        L = _C.lib()
        cam_c, j2d, conf = _f32(const['camera_center']), _f32(const['joints_2d']), _f32(const['joints_conf'])
        means, precisions, logw = const['means'], const['precisions'], const['log_weights']
        small = torch.empty(b, 2, dtype=torch.float32, device=v.device)
        gj = torch.empty_like(j)
        gc = torch.empty(b, 3, dtype=torch.float32, device=v.device)
        gp = torch.empty(b, 69, dtype=torch.float32, device=v.device)
        p = model.num_pairs if select is not None else 0

        # ONE cleared buffer: the vertex gradient the unit backward accumulates into, the arrival counter of the
        # tail kernel ("the last block adds up": it belongs to THIS call -- no counter shared between streams, none
        # left non-zero by an aborted launch) and the region pairs' keys (tuch_region_pair_keys wants them zero).  It is
        # cleared by the FIRST kernel of the search (a fill launch of its own was 5 us of the second stream's chain).
        n = v.numel()
        k0 = n + 1 + ((n + 1) & 1)                  # the 64-bit keys start on an even word behind the counter
        det = 2 * n if deterministic() else 0       # deterministic mode: 64-bit fixed-point accumulators

        def beside_the_walk(zeros):          # on the second stream (ContactModel.exterior_and_partner), behind the search
            keys = zeros[k0:k0 + 2 * b * p].view(torch.int64) if p else None
            fixed = zeros[k0 + 2 * b * p:k0 + 2 * b * p + det].view(torch.int64) if det else None
            if p:
                _C.check(L.tuch_region_pair_keys(model._handle, _C.ptr(v), b, _C.ptr(select), 1, _C.ptr(keys), _C.stream()))
            _C.check(L.tuch_smplify_small_terms(
                _C.ptr(j), _C.ptr(cam_t), _C.ptr(cam_c), _C.ptr(j2d), _C.ptr(conf), _C.ptr(pose), _C.ptr(means),
                _C.ptr(precisions), _C.ptr(logw), b, nj, means.shape[0], float(const['focal']), float(const['sigma']),
                float(const['prior_scale']), _C.ptr(small), _C.ptr(gj), _C.ptr(gc), _C.ptr(gp), _C.stream()))
            return (keys, fixed, small, gj, gc, gp, zeros[:n].view(v.shape), zeros[n:n + 1].view(torch.int32))
        # (cap: the contact term of losses.py:96-105 looks at the partner of an exterior vertex only within euclthres; 0.1 % more,
        # because the term recomputes the distance with its own rounding)
        exterior, _, partner, _extra = model.exterior_and_partner(v, apply_segments=const['apply_segments'],
                                                                  also=beside_the_walk, zero_floats=k0 + 2 * b * p + det,
                                                                  iterative=True,       # SMPLify-DC's stage-2 loop
                                                                  cap=1.001 * max(float(const['euclthres']), 0.0) + 1e-6)
        out = torch.empty(1, dtype=torch.float32, device=v.device)
        share = torch.empty(L.tuch_smplify_stage2_fused_scratch_floats(b), dtype=torch.float32, device=v.device)
        # the objective is the root of the fit's graph: its vertex gradient for a unit upstream gradient is written by the
        # same launch that forms the sums (csrc/contact_terms.hip: stage2_fused_kernel); backward() only hands it over
        want_grad = any(ctx.needs_input_grad[:4])
        gv = _extra[6] if want_grad else None
        # deterministic mode: the vertex gradient stays in its 64-bit fixed-point accumulators (no conversion launch here);
        # backward() hands them to the body model's node, whose skinning adjoint reads them -- or converts them if the pass
        # is not the plain `objective.backward()` of a fit.  gv (zeros nobody writes in this mode) is what autograd carries.
        fixed = _extra[1] if want_grad else None
        _C.check(L.tuch_smplify_stage2_fused(_C.ptr(v), _C.ptr(partner), _C.ptr(exterior), _C.ptr(valid), b, v.shape[1],
                                             MODE_SMPLIFY, float(const['euclthres']), _C.ptr(small), None, None, p,
                                             float(const['contact_scale']), float(const['r2r_scale']), _C.ptr(share),
                                             _C.ptr(_extra[7]), None, _C.ptr(out), _C.ptr(gv) if fixed is None else None,
                                             model._handle if p else None, _C.ptr(_extra[0]),
                                             _C.ptr(fixed), _C.stream()))
        ctx.fixed = fixed
        if want_grad:
            ctx.save_for_backward(gv, gj, gc, gp)
        ctx.in_dtypes = (verts.dtype, joints.dtype, camera_t.dtype, body_pose.dtype)
        # body_pose also feeds the body model that made `verts`: that node's backward runs after this one (it waits for the
        # vertex gradient) and can add the prior's pose gradient inside its own last kernel (lbs._SmplLBS: pose_grad_extra)
        # instead of autograd summing two gradients in a launch of its own
        node = verts.grad_fn
        # (somebody who retains / hooks the vertices' own gradient must see the real numbers, not the carrier of zeros)
        ctx.verts_unwatched = not verts.retains_grad and not getattr(verts, '_backward_hooks', None)
        ref = getattr(node, 'pose_ref', None) if node is not None else None
        ctx.lbs_node = node if (want_grad and ctx.needs_input_grad[0] and ctx.needs_input_grad[3] and ref is not None
                                and ref() is body_pose) else None
        return out[0]

    @staticmethod
    @once_differentiable          # the gradients were formed in forward(): constants of a second differentiation
    def backward(ctx, g):
        gv, gj, gc, gp = ctx.saved_tensors
        dv, dj, dc, dp = ctx.in_dtypes
        # loss.backward() through ops.backward_scalar seeds the graph with a cached tensor of ones: recognised by its
        # address (no device round trip).  Any other upstream gradient scales the unit gradients.
        unit = any(g.data_ptr() == seed.data_ptr() for seed in _ONES.values())
        fixed, ctx.fixed = ctx.fixed, None
        hand_over = (fixed is not None and unit and ctx.lbs_node is not None and _graph_task_id() >= 0
                     and is_root_of_backward_scalar(ctx) and ctx.verts_unwatched)
        if fixed is not None and not hand_over:
            # anything but the plain backward of a fit (a scaled upstream gradient, a gradient asked for the vertices
            # themselves, a loss with further terms): the float gradient, by a conversion launch
            gv = torch.empty_like(gv)
            _C.check(_C.lib().tuch_fixed_to_float(_C.ptr(fixed), fixed.numel(), _C.ptr(gv), _C.stream()))
        if not unit:
            g = g.reshape(()).to(torch.float32)
            gv, gj, gc, gp = gv * g, gj * g, gc * g, gp * g
        # the ROOT of the pass: backward_scalar() was called on THIS node's output (a loss like `tail + other(body_pose)`
        # reaches this node with the same unit seed -- AddBackward forwards it unchanged -- but then not every gradient of
        # the parameters flows through here, and the optimiser's update must not be applied from this node's gradient alone)
        root = unit and is_root_of_backward_scalar(ctx)
        if ctx.lbs_node is not None and _graph_task_id() >= 0:
            # tagged with THIS backward pass: the body model's node takes it only within the same pass (a gradient left by
            # a pass that never reached that node must not leak into a later one)
            ctx.lbs_node.pose_grad_extra = (gp, _graph_task_id())
            if hand_over:
                ctx.lbs_node.verts_grad_fixed = (fixed, _graph_task_id())
            # this node is the ROOT of the pass (see above): every gradient of the fit's parameters flows through what it
            # returns -- the body model's node may then apply the optimiser's update itself (lbs.py)
            ctx.lbs_node.root_pass = _graph_task_id() if root else None
            return gv.to(dv), gj.to(dj), gc.to(dc), None, None, None, None, None
        return gv.to(dv), gj.to(dj), gc.to(dc), gp.to(dp), None, None, None, None


def smplify_stage2_tail(verts, joints, camera_t, body_pose, model, valid_u8, select_u8, **const):
    """Scalar stage-2 objective behind the body model (see _Stage2Tail); const: camera_center, joints_2d, joints_conf,
    means, precisions, log_weights, focal, sigma, prior_scale, euclthres, contact_scale, r2r_scale, apply_segments."""
    return _Stage2Tail.apply(verts, joints, camera_t, body_pose, model, valid_u8, select_u8, const)


_DERIVED = {}


def cached_derived(key_tensors, build):
    """Small per-call tensors derived from caller inputs that do not change between iterations
    (validity / selection masks): rebuilt only when an input was modified in place or replaced.  The
    entry holds its source tensors, so their addresses cannot be reused while it is cached.  A captured
    hipGraph holds the address of the derived tensor: entries are kept for 256 distinct inputs before the oldest
    is dropped, far more than the graphs a process keeps alive at once."""
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) if t is not None else None for t in key_tensors)
    hit = _DERIVED.get(key)
    if hit is None:
        while len(_DERIVED) >= 256:
            _DERIVED.pop(next(iter(_DERIVED)))
        hit = (build(), key_tensors)
        _DERIVED[key] = hit
    return hit[0]


# ------------------------------------------------------------------------- model
def region_tables(cdict):
    """{'classes': [(regA, regB), ...], 'csig': {region: vertex ids}} (train_module.py:64-66) ->
    (ordered vertex-id lists, pairs [P,2] of indices into that order).  Region keys are looked up as the
    reference does (``csig[regpair[0]]``), whatever their type; numpy string scalars match their str()."""
    csig = cdict['csig']
    names = list(csig.keys())
    index = {n: i for i, n in enumerate(names)}

    def find(key):
        for k in (key, str(key)):
            try:
                if k in index:
                    return index[k]
            except TypeError:
                pass
        if hasattr(key, 'item'):
            return find(key.item())
        raise KeyError('region %r of cdict[\'classes\'] is not a key of cdict[\'csig\']' % (key,))
    regions = [np.asarray(csig[n], dtype=np.int64) for n in names]
    pairs = np.asarray([[find(p[0]), find(p[1])] for p in cdict['classes']], np.int64).reshape(-1, 2)
    return regions, pairs


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def segment_faces(body_faces: np.ndarray, vidx: np.ndarray, bands: Sequence[np.ndarray], first_cap: int,
                  num_verts: Optional[int] = None) -> np.ndarray:
    """Closed-segment face list of tuch/utils/segmentation.py:48-66: body faces whose three
    vertices are all in ``vidx`` followed by the cap fans [b[i+1], b[i], cap_k]; cap k of this
    segment is stored at global vertex index ``first_cap + k``."""
    body_faces = np.asarray(body_faces, np.int64)
    inside = np.isin(body_faces, np.asarray(vidx)).sum(1) == 3
    fans = []
    for k, band in enumerate(bands):
        band = np.asarray(band, np.int64)
        fans.append(np.stack([band[1:], band[:-1], np.full(len(band) - 1, first_cap + k, np.int64)], 1))
    return np.concatenate([body_faces[inside]] + fans, 0) if fans else body_faces[inside]


def cluster_tree(faces, num_verts: int, leaf_faces: int = 64) -> dict:
    """The face-cluster tree used by the hierarchical winding numbers (host only; see include/tuch_amd.h,
    tuch_cluster_tree_build).  Returns numpy arrays: nodes [N,8], vidx, sign, qperm, frontier_off,
    frontier_nodes, launch_order, rows, face_leaf, plus exact_len."""
    L = _C.lib()
    f = np.ascontiguousarray(np.asarray(faces).reshape(-1, 3), dtype=np.int32)
    h = ctypes.c_void_p()
    _C.check(L.tuch_cluster_tree_build(int(num_verts), f.shape[0], f.ctypes.data_as(ctypes.c_void_p), int(leaf_faces),
                                       ctypes.byref(h)))
    try:
        vals = [ctypes.c_int(0) for _ in range(6)]
        _C.check(L.tuch_cluster_tree_info(h, *[ctypes.byref(v) for v in vals]))
        n, exact_len, stream_len, qblocks, nfr, frtot = [v.value for v in vals]
        out = dict(nodes=np.zeros((n, 8), np.int32), vidx=np.zeros(stream_len, np.int32),
                   sign=np.zeros(stream_len, np.float32), qperm=np.zeros(qblocks * 128, np.int32),
                   frontier_off=np.zeros(nfr + 1, np.int32), frontier_nodes=np.zeros(frtot, np.int32),
                   launch_order=np.zeros(frtot * qblocks, np.int32), rows=np.zeros((n, 2), np.int32),
                   face_leaf=np.zeros(f.shape[0], np.int32))
        _C.check(L.tuch_cluster_tree_export(h, *[out[k].ctypes.data_as(ctypes.c_void_p) for k in
                                                 ('nodes', 'vidx', 'sign', 'qperm', 'frontier_off', 'frontier_nodes',
                                                  'launch_order', 'rows', 'face_leaf')]))
        out['exact_len'] = exact_len
        return out
    finally:
        L.tuch_cluster_tree_free(h)


class ContactModel:
    """Device-side constants of one body model; wraps tuch_contact_model.

    faces      [F,3] ints                       (smpl.faces, train.py:62)
    geomask    [V,V] bool, geod > geothres      (smplifydc.py:65, loss.py:71), optional
    segments   list of (vidx, [band loops])     (segmentation.py), optional
    regions    ordered list of vertex-id lists; pairs [P,2] region indices, optional
    """

    def __init__(self, faces, geomask=None, segments=None, regions=None, pairs=None,
                 device: Optional[torch.device] = None):
        self.device = torch.device(device if device is not None else 'cuda')
        faces = np.asarray(faces.detach().cpu() if isinstance(faces, torch.Tensor) else faces)
        self.faces_np = faces.astype(np.int64)
        self._hints = {}
        self._prev_flags = {}
        self.num_verts = int(faces.max()) + 1
        self.num_faces = int(faces.shape[0])
        gm = None
        if geomask is not None:
            gm = geomask.detach().cpu().numpy() if isinstance(geomask, torch.Tensor) else np.asarray(geomask)
            self.num_verts = max(self.num_verts, gm.shape[0])
            gm = np.ascontiguousarray(gm.astype(np.uint8))
            assert gm.shape == (self.num_verts, self.num_verts)
        self.has_mask = gm is not None
        v = self.num_verts
        # segment tables
        segments = list(segments or [])
        seg_q_off, seg_q, seg_f_off, seg_f, cap_off, cap_v = [0], [], [0], [], [0], []
        for vidx, bands in segments:
            first_cap = v + len(cap_off) - 1
            seg_f.append(segment_faces(self.faces_np, vidx, bands, first_cap))
            seg_f_off.append(seg_f_off[-1] + len(seg_f[-1]))
            seg_q.append(np.asarray(vidx, np.int64))
            seg_q_off.append(seg_q_off[-1] + len(vidx))
            for band in bands:
                cap_v.append(np.asarray(band, np.int64))
                cap_off.append(cap_off[-1] + len(band))
        self.num_segments = len(segments)
        self.seg_q_total = seg_q_off[-1]
        self.seg_q_off = seg_q_off
        self.seg_vidx = [np.asarray(s[0], np.int64) for s in segments]
        # region tables
        regions = list(regions or [])
        reg_off = np.cumsum([0] + [len(r) for r in regions])
        self.num_pairs = 0 if pairs is None else len(pairs)
        cat = lambda lst, width=None: (np.concatenate(lst) if lst else np.zeros(0, np.int64))
        keep = dict(
            faces=_i32(faces), seg_q_off=_i32(seg_q_off), seg_q=_i32(cat(seg_q)), seg_f_off=_i32(seg_f_off),
            seg_f=_i32(cat(seg_f).reshape(-1, 3)), cap_off=_i32(cap_off), cap_v=_i32(cat(cap_v)),
            reg_off=_i32(reg_off), reg_v=_i32(cat([np.asarray(r) for r in regions])),
            pairs=_i32(pairs if pairs is not None else np.zeros((0, 2))))
        # the device copy (tuch_contact_model_create: the library's only allocations) is made on first use, so
        # that the callers' constructors (RegressorLoss, SMPLifyDC, ...) also run where no GPU is visible
        self._host = (keep, gm, len(cap_off) - 1, len(regions))
        self._h = None
        self._faces_i32 = None
        # host-side switches, read from the environment once (like the library's own, tuch_contact_model_set_option):
        # overlap = search on a second stream beside the inside test, v2v_hint = partner hints kept between calls,
        # inside_first = the inside test's chain is enqueued before the search
        self._py_options = {'overlap': int(os.environ.get('TUCH_OVERLAP', '1') != '0'),
                            'v2v_hint': int(os.environ.get('TUCH_V2V_HINT', '1') != '0'),
                            'inside_first': int(os.environ.get('TUCH_INSIDE_FIRST', '1') != '0')}
        self._pending_options = {}

    @property
    def _handle(self):
        if self._h is None:
            keep, gm, num_caps, num_regions = self._host
            host_tables = os.environ.get('TUCH_HOST_TABLES', '0') not in ('', '0')    # sanitizer runs of the table builders
            if self.device.type != 'cuda' and not host_tables:
                raise _C.TuchError('tuch_amd kernels need a HIP device, the model was created for %s' % self.device)
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else ctypes.c_void_p(0)
            handle = ctypes.c_void_p(0)
            with (contextlib.nullcontext() if host_tables else torch.cuda.device(self.device)):
                _C.check(_C.lib().tuch_contact_model_create(
                    ctypes.byref(handle), self.num_verts, self.num_faces, p(keep['faces']),
                    gm.ctypes.data_as(ctypes.c_void_p) if gm is not None else ctypes.c_void_p(0),
                    self.num_segments, p(keep['seg_q_off']), p(keep['seg_q']), p(keep['seg_f_off']), p(keep['seg_f']),
                    num_caps, p(keep['cap_off']), p(keep['cap_v']),
                    num_regions, p(keep['reg_off']), p(keep['reg_v']), self.num_pairs, p(keep['pairs'])))
            self._h = handle
            self._host = None          # the 47 MB byte mask is not needed again
            for name, value in self._pending_options.items():
                _C.check(_C.lib().tuch_contact_model_set_option(handle, name.encode(), int(value)))
            self._pending_options = {}
        return self._h

    def set_option(self, name: str, value: int) -> None:
        """Change one of the model's switches (include/tuch_amd.h: tuch_contact_model_set_option; plus the host-side
        'overlap' and 'v2v_hint').  The environment is read once, when the model is created; hot calls never read it.
        Captured hipGraphs keep the behaviour they were captured with."""
        if name in self._py_options:
            self._py_options[name] = int(value)
        elif self._h is None:
            self._pending_options[name] = int(value)
        else:
            _C.check(_C.lib().tuch_contact_model_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        if name in self._py_options:
            return self._py_options[name]
        out = ctypes.c_int(0)
        _C.check(_C.lib().tuch_contact_model_get_option(self._handle, name.encode(), ctypes.byref(out)))
        return out.value

    def canary_hits(self, reset: bool = True) -> int:
        """Guard words between workspace regions found changed since the last reset (option canary = 1: debug mode,
        include/tuch_amd.h).  Synchronises the device."""
        out = ctypes.c_int(0)
        _C.check(_C.lib().tuch_contact_model_canary_hits(self._handle, ctypes.byref(out), int(reset)))
        return out.value

    def canary_selftest(self) -> int:
        """Overruns a guarded region on purpose: returns the hits counted (1 if the mechanism works)."""
        ws = _workspace(8192, self.device)
        rc = _C.lib().tuch_contact_model_canary_selftest(self._handle, _C.ptr(ws), ws.numel(), _C.stream())
        if rc < 0:
            _C.check(rc)
        return rc

    @property
    def faces_i32(self):
        if self._faces_i32 is None:
            self._faces_i32 = torch.as_tensor(_i32(self.faces_np), device=self.device)
        return self._faces_i32

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _C.lib().tuch_contact_model_destroy(h)
            except Exception:
                pass
            self._h = None

    def strips(self):
        """(vertex ids, signs, number of strips) of the triangle-strip walk used by the winding kernel."""
        n, k = ctypes.c_int(0), ctypes.c_int(0)
        L = _C.lib()
        _C.check(L.tuch_contact_model_strips(self._handle, ctypes.byref(n), ctypes.byref(k), None, None))
        vidx = np.zeros(n.value, np.int32)
        sign = np.zeros(n.value, np.float32)
        _C.check(L.tuch_contact_model_strips(self._handle, None, None, vidx.ctypes.data_as(ctypes.c_void_p),
                                             sign.ctypes.data_as(ctypes.c_void_p)))
        return vidx, sign, k.value

    def exterior_and_partner(self, verts: torch.Tensor, apply_segments: bool = True, also=None, zero_floats: int = 0,
                             iterative: bool = False, cap: Optional[float] = None):
        """exterior_flags + v2v_min of the same vertices -> (exterior, min_d2, partner[, also()]).
        The two only share their input: the nearest-vertex search (and the optional callable ``also``,
        e.g. the region pairs) runs on a second stream so that its tail fills the gaps of the long winding
        walk (option overlap = 0 keeps everything on the current stream).
        zero_floats > 0: a float32 buffer of (at least) that many ZEROS is handed to ``also(buffer)`` -- cleared by the
        search's first kernel (tuch_v2v_min_model_shared_zero), not by a fill launch.
        iterative: the caller is an iterative fit (the previous call's partners, kept as hints, are almost this call's):
        the search then uses fewer, longer wavefronts (see v2v_min).  Results never depend on it.
        cap: the caller only needs the partners of vertices that are INSIDE the body or have one within ``cap`` (the SMPLify-DC
        contact term, tuch/smplify/losses.py:96-105, with cap >= euclthres): the search of the vertices the previous call found
        outside starts at the cap (half the search's time at the bench's batch; the step does not follow: option v2v_cap, off), and the few that this call's inside test finds
        inside after all are searched again, exhaustively, behind the inside test (v2v_fix).  For every other vertex min_d2
        is cap^2 and the partner some admissible vertex farther than cap.  Exact for what the caller uses; see
        include/tuch_amd.h: tuch_v2v_min_model_capped."""
        zero = None
        if zero_floats > 0:
            zero = torch.empty((int(zero_floats) + 3) // 4 * 4, dtype=torch.float32, device=verts.device)
            call_also = (lambda: also(zero)) if also is not None else None
        else:
            call_also = also
        capped = cap is not None and verts.is_cuda and self.v2v_can_cap(verts.shape[0])
        if not (verts.is_cuda and self._py_options['overlap']):
            exterior = self.exterior_flags(verts, apply_segments=apply_segments)
            if capped:
                mn, partner, state = self.v2v_min(verts, zero=zero, iterative=iterative, cap=cap)
                self.v2v_fix(exterior, mn, partner, state)
            else:
                mn, partner = self.v2v_min(verts, zero=zero, iterative=iterative)
            return exterior, mn, partner, (call_also() if call_also is not None else None)
        cur = _current_stream(verts.device)
        side = _side_stream(verts.device)
        side.wait_stream(cur)
        if capped:
            v = _f32(verts)
            exterior = self.exterior_flags(v, apply_segments=apply_segments)     # (the inside test's chain first, as below)
            with torch.cuda.stream(side):
                if zero is not None:
                    zero.record_stream(side)
                mn, partner, state = self.v2v_min(v, leave_room=True, zero=zero, iterative=iterative, cap=cap)
                extra = call_also() if call_also is not None else None
            cur.wait_stream(side)
            for t in (mn, partner, state[0]) + (tuple(extra) if isinstance(extra, (tuple, list)) else (extra,)):
                if torch.is_tensor(t):
                    t.record_stream(cur)
            self.v2v_fix(exterior, mn, partner, state)       # behind both: the final flags, the capped search's keys
            return exterior, mn, partner, extra
        first = self._py_options.get('inside_first', 0)
        if first:
            # the inside test's chain FIRST (the side stream waits only for what was enqueued before its wait above).
            # Launched -- or captured -- behind the search, the chain's small head kernels find every wave slot taken by
            # the search's 55 k one-wave workgroups and only get going when it is done (eager: ray_leaf_bounds 119 us
            # instead of 14; replayed graph at batch 8: 0.240 against 0.213 ms once tree_inner_bounds_kernel no longer
            # delays the search's start)
            exterior = self.exterior_flags(verts, apply_segments=apply_segments)
        with torch.cuda.stream(side):
            if zero is not None:
                zero.record_stream(side)
            mn, partner = self.v2v_min(verts, leave_room=True, zero=zero, iterative=iterative)
            # (the caller's extra work -- region pairs, reprojection + prior -- FIRST, beside the short head of the inside
            # test's chain, was measured: 0.587 against 0.552 ms per step; it delays the search, which the chain waits for)
            extra = call_also() if call_also is not None else None
        if not first:
            exterior = self.exterior_flags(verts, apply_segments=apply_segments)
        cur.wait_stream(side)
        for t in (mn, partner) + (tuple(extra) if isinstance(extra, (tuple, list)) else (extra,)):
            if torch.is_tensor(t):
                t.record_stream(cur)
        return exterior, mn, partner, extra

    def winding_tree_work(self, verts: torch.Tensor) -> dict:
        """Stream elements the hierarchical winding walk steps through for these vertices (measurement aid)."""
        verts = _f32(verts)
        b = verts.shape[0]
        L = _C.lib()
        nbytes = L.tuch_exterior_workspace_bytes(self._handle, b)
        ws = _workspace(nbytes, verts.device)
        out = (ctypes.c_ulonglong * 4)()
        _C.check(L.tuch_winding_tree_work(self._handle, _C.ptr(verts), b, _C.ptr(ws), nbytes, out, _C.stream()))
        return dict(leaf_elements=int(out[0]), cap_elements=int(out[1]), wavefronts=int(out[2]),
                    flat_stream_elements=int(out[3]), queries_per_step=64,
                    query_blocks=-(-self.num_verts // 128) * 2 * b)

    def ray_work(self, verts: torch.Tensor) -> dict:
        """Strip elements the ray-crossing inside test steps through for these vertices (measurement aid)."""
        verts = _f32(verts)
        b = verts.shape[0]
        L = _C.lib()
        nbytes = L.tuch_exterior_workspace_bytes(self._handle, b)
        ws = _workspace(nbytes, verts.device)
        out = (ctypes.c_ulonglong * 4)()
        _C.check(L.tuch_ray_work(self._handle, _C.ptr(verts), b, _C.ptr(ws), nbytes, out, _C.stream()))
        return dict(elements=int(out[0]), wavefronts=int(out[3]), queries_per_step=64,
                    lanes_inside_leaf_slabs=int(out[1]) / max(int(out[2]), 1))

    # K2 + K3
    def exterior_flags(self, verts: torch.Tensor, apply_segments: bool = True, thresh: float = 0.99,
                       return_details: bool = False):
        verts = _f32(verts)
        b = verts.shape[0]
        assert verts.shape[1] == self.num_verts
        L = _C.lib()
        ext = torch.empty(b, self.num_verts, dtype=torch.uint8, device=verts.device)
        w = torch.empty(b, self.num_verts, dtype=torch.float32, device=verts.device) if return_details else None
        seg_w = seg_e = None
        if return_details and self.num_segments:
            seg_w = torch.empty(b, self.seg_q_total, dtype=torch.float32, device=verts.device)
            seg_e = torch.empty(b, self.seg_q_total, dtype=torch.uint8, device=verts.device)
        nbytes = L.tuch_exterior_workspace_bytes(self._handle, b)
        ws = _workspace(nbytes, verts.device)
        _C.check(L.tuch_exterior_flags(self._handle, _C.ptr(verts), b, int(apply_segments), float(thresh),
                                       _C.ptr(w), _C.ptr(ext), _C.ptr(seg_w), _C.ptr(seg_e), _C.ptr(ws), nbytes,
                                       _C.stream()))
        return (ext, w, seg_w, seg_e) if return_details else ext

    # K1
    def v2v_min(self, verts: torch.Tensor, leave_room: bool = False, zero: Optional[torch.Tensor] = None, iterative: bool = False,
                cap: Optional[float] = None):
        """leave_room: other kernels run beside the search on another stream (tuch_v2v_min_model_shared).
        iterative: the hints are expected to be near-final (SMPLify-DC's loops): a quarter of the wavefronts, each over more
        leaves -- faster with good bounds (0.412 against 0.426 ms per stage-2 step at batch 64), slower on new bodies.
        zero: a caller tensor (numel * itemsize a multiple of 16) cleared by the call's first kernel (..._shared_zero)."""
        if not self.has_mask:
            raise _C.TuchError('ContactModel was created without a geodesic mask')
        verts = _f32(verts)
        b = verts.shape[0]
        assert verts.shape[1] == self.num_verts
        L = _C.lib()
        mn = torch.empty(b, self.num_verts, dtype=torch.float32, device=verts.device)
        arg = torch.empty(b, self.num_verts, dtype=torch.int32, device=verts.device)
        nbytes = L.tuch_v2v_model_workspace_bytes(self._handle, b)
        ws = _workspace(nbytes, verts.device)
        flags = int(bool(leave_room)) | (2 if iterative else 0)
        zbytes = zero.numel() * zero.element_size() if zero is not None else 0
        if cap is not None:
            # the capped form (see v2v_capped): the caller finishes it with v2v_fix(...) on the returned state
            prev = self._v2v_prev(b)
            _C.check(L.tuch_v2v_min_model_capped(self._handle, _C.ptr(verts), b, _C.ptr(mn), _C.ptr(arg), _C.ptr(self._v2v_hint(b)),
                                                 _C.ptr(ws), nbytes, flags, _C.ptr(zero), zbytes, _C.ptr(prev), float(cap), _C.stream()))
            return mn, arg, (ws, nbytes, prev, float(cap))
        _C.check(L.tuch_v2v_min_model_shared_zero(self._handle, _C.ptr(verts), b, _C.ptr(mn), _C.ptr(arg),
                                                  _C.ptr(self._v2v_hint(b)), _C.ptr(ws), nbytes, flags, _C.ptr(zero), zbytes, _C.stream()))
        return mn, arg

    def v2v_can_cap(self, batch: int) -> bool:
        """Can the search of this model run capped (tuch_v2v_min_model_capped: the leaf scan with hints, option v2v_cap)?"""
        return bool(self.has_mask and _C.lib().tuch_v2v_min_model_can_cap(self._handle)) and self._v2v_hint(batch) is not None

    def v2v_fix(self, exterior: torch.Tensor, mn: torch.Tensor, arg: torch.Tensor, state) -> None:
        """Second half of a capped search: the vertices that were cut off at the cap and that ``exterior`` [B,V] u8 (this
        call's inside test) shows inside are searched exhaustively; mn / arg
        are corrected in place, the flags become the next call's prediction."""
        ws, nbytes, prev, cap = state
        b = exterior.shape[0]
        _C.check(_C.lib().tuch_v2v_min_model_fix(self._handle, b, _C.ptr(exterior), _C.ptr(prev), cap, _C.ptr(mn), _C.ptr(arg),
                                                 _C.ptr(self._v2v_hint(b)), _C.ptr(ws), nbytes, _C.stream()))

    def _v2v_prev(self, batch: int) -> torch.Tensor:
        """Persistent per-batch-size flags of the previous capped call ([B,V] u8, 1 = outside the body; zeros at first:
        nothing is capped).  A prediction only -- results never depend on it."""
        buf = self._prev_flags.get(batch)
        if buf is None:
            buf = self._prev_flags[batch] = torch.zeros(batch, self.num_verts, dtype=torch.uint8, device=self.device)
        return buf

    def _v2v_hint(self, batch: int) -> Optional[torch.Tensor]:
        """Persistent per-batch-size buffer in which the search leaves its partners for the next call (an iterative
        fit re-finds almost the same partners): seeds only, the results never depend on it (option v2v_hint = 0: off).
        Buffers are kept for the life of the model (a captured hipGraph may hold their addresses)."""
        if not self._py_options['v2v_hint']:
            return None
        buf = self._hints.get(batch)
        if buf is None:
            n = _C.lib().tuch_v2v_hint_bytes(self._handle, batch)
            if n == 0:
                return None
            buf = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._hints[batch] = buf
        return buf

    def winding_points(self, verts: torch.Tensor, points: torch.Tensor, counts: Optional[torch.Tensor] = None,
                       thresh: float = 0.99, flags_only: bool = False):
        """Winding numbers of arbitrary points [B,Q,3] against this mesh posed by verts [B,V,3]
        (cluster-tree walk; flat triangle strips for meshes without a tree); counts [B] int32 marks how many points per
        body are real.  Any point order is exact; blocks of 64 consecutive points that are close in space are fast.
        flags_only: return (None, exterior); for points OFF the surface the winding number is an integer and the flags
        come from signed ray crossings (csrc/ray_winding.hip) instead of the solid-angle sum."""
        v, pts = _f32(verts), _f32(points)
        b, q, _ = pts.shape
        L = _C.lib()
        w = torch.empty(b, q, dtype=torch.float32, device=pts.device) if not flags_only else None
        ext = torch.empty(b, q, dtype=torch.uint8, device=pts.device)
        nbytes = L.tuch_winding_points_workspace_bytes(self._handle, b, q)
        ws = _workspace(nbytes, pts.device)
        _C.check(L.tuch_winding_points(self._handle, _C.ptr(v), _C.ptr(pts), _C.ptr(counts), b, q, float(thresh),
                                       _C.ptr(w), _C.ptr(ext), _C.ptr(ws), nbytes, _C.stream()))
        return w, ext

    def v2v_min_indexed(self, points: torch.Tensor, vertex_ids: torch.Tensor, offsets: torch.Tensor,
                        max_points: int, tree_order: bool = False, mfma: bool = False):
        """Ragged masked nearest neighbour (HD points, loss.py:288-291).  points [N,3], vertex_ids [N]
        int32, offsets [B+1] int32 -> (min_d2 [N], argmin [N] int32 relative to the body's first point).
        tree_order: vertex_ids are positions in the cluster tree's vertex order (``tree_positions``) and the
        mask packed in that order is used -- the same answer, far fewer distinct mask words per wavefront.
        mfma: the matrix-core form the HD branch runs (tuch_v2v_min_indexed_mfma: winners may differ between rows that tie
        within ~1e-6 relative)."""
        if not self.has_mask:
            raise _C.TuchError('ContactModel was created without a geodesic mask')
        pts = _f32(points)
        n = pts.shape[0]
        mn = torch.empty(n, dtype=torch.float32, device=pts.device)
        arg = torch.empty(n, dtype=torch.int32, device=pts.device)
        L = _C.lib()
        bits = L.tuch_contact_model_tree_mask_bits(self._handle) if tree_order else \
            L.tuch_contact_model_mask_bits(self._handle)
        if not bits:
            raise _C.TuchError('ContactModel has no mask in tree order (no cluster tree)')
        size_fn, fn = (L.tuch_v2v_min_indexed_mfma_workspace_bytes, L.tuch_v2v_min_indexed_mfma) if mfma else \
            (L.tuch_v2v_min_indexed_workspace_bytes, L.tuch_v2v_min_indexed)
        nbytes = size_fn(offsets.shape[0] - 1, int(max_points))
        ws = _workspace(nbytes, pts.device)
        vertex_ids, offsets = vertex_ids.contiguous(), offsets.contiguous()
        _C.check(fn(_C.ptr(pts), _C.ptr(vertex_ids), _C.ptr(offsets),
                    ctypes.c_void_p(bits), offsets.shape[0] - 1, self.num_verts, int(max_points),
                    _C.ptr(mn), _C.ptr(arg), _C.ptr(ws), nbytes, _C.stream()))
        return mn, arg

    def tree_positions(self) -> Optional[np.ndarray]:
        """position of every vertex in the cluster tree's vertex order (inverse of the MODEL'S OWN qperm, whatever leaf
        size it was built with: tuch_contact_model_tree_order), or None without tree."""
        qperm = np.zeros(self.num_verts, np.int32)
        try:
            _C.check(_C.lib().tuch_contact_model_tree_order(self._handle, qperm.ctypes.data_as(ctypes.c_void_p), None))
        except _C.TuchError:
            return None
        pos = np.empty(self.num_verts, np.int32)
        pos[qperm] = np.arange(self.num_verts, dtype=np.int32)
        return pos

    # K5
    def region_pair_min(self, verts: torch.Tensor, select: Optional[torch.Tensor] = None, masked: bool = False):
        return _RegionPairMin.apply(verts, self, select, masked)


class HDModel:
    """Device tables of the HD vertex regressor for the fused HD branch (wraps tuch_hd_model; csrc/hd_contact.hip).
    hd_idx / hd_w [N,K]: the K <= 8 non-zeros of every regressor row (K = 3: barycentric samples); hd_face [N]:
    faces_vert_is_sampled_from."""

    def __init__(self, contact_model: ContactModel, hd_idx, hd_w, hd_face):
        self.contact_model = contact_model
        self.num_points = int(np.asarray(hd_face).shape[0])
        hd_idx, hd_w = np.asarray(hd_idx), np.asarray(hd_w)
        # [N,K]: K non-zeros per regressor row (3 for barycentric samples; up to 8)
        self.row_nnz = int(hd_idx.shape[-1]) if hd_idx.ndim == 2 else 3
        if not 1 <= self.row_nnz <= 8 or hd_idx.size != self.num_points * self.row_nnz or hd_w.size != hd_idx.size:
            raise ValueError('HDModel: hd_idx / hd_w must be [N,K] with 1 <= K <= 8 (got %s, %s for %d points)'
                             % (hd_idx.shape, hd_w.shape, self.num_points))
        self._host = (_i32(hd_idx.reshape(-1, self.row_nnz)), np.ascontiguousarray(hd_w.reshape(-1, self.row_nnz), np.float32),
                      _i32(hd_face))
        self._h = None

    @property
    def _handle(self):
        if self._h is None:
            idx, w, face = self._host
            handle = ctypes.c_void_p(0)
            cm = self.contact_model._handle
            host_tables = os.environ.get('TUCH_HOST_TABLES', '0') not in ('', '0')
            with (contextlib.nullcontext() if host_tables else torch.cuda.device(self.contact_model.device)):
                _C.check(_C.lib().tuch_hd_model_create_k(ctypes.byref(handle), cm, self.num_points, self.row_nnz,
                                                         idx.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p),
                                                         face.ctypes.data_as(ctypes.c_void_p)))
            self._h = handle
        return self._h

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _C.lib().tuch_hd_model_destroy(h)
            except Exception:
                pass
            self._h = None

    def contact_terms(self, verts, exterior, min_d2, partner, valid_u8, euclthres, thresh=0.99):
        """[B,2] (interior sum, exterior sum) over the HD points of loss.py:274-315, differentiable w.r.t. verts."""
        return _HDContact.apply(verts, exterior, min_d2, partner, valid_u8, self, float(euclthres), float(thresh))

    def selection(self, saved, batch):
        """(counts [B], selected [B,N] caller-order ids per slot, -1 padded) of a forward call (tests)."""
        counts = np.zeros(batch, np.int32)
        sel = np.zeros((batch, self.num_points), np.int32)
        _C.check(_C.lib().tuch_hd_contact_selection(self._handle, _C.ptr(saved), batch, counts.ctypes.data_as(ctypes.c_void_p),
                                                    sel.ctypes.data_as(ctypes.c_void_p)))
        return counts, sel


    def details(self, saved, batch):
        """(partner [B,N] caller-order id of every slot's partner or -1, exterior [B,N] bool) of a forward call (tests)."""
        part = np.zeros((batch, self.num_points), np.int32)
        ext = np.zeros((batch, self.num_points), np.uint8)
        _C.check(_C.lib().tuch_hd_contact_details(self._handle, _C.ptr(saved), batch, part.ctypes.data_as(ctypes.c_void_p),
                                                  ext.ctypes.data_as(ctypes.c_void_p)))
        return part, ext.astype(bool)


class _HDContact(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, exterior, min_d2, partner, valid, hm: HDModel, euclthres, thresh):
        v = _f32(verts)
        b = v.shape[0]
        L = _C.lib()
        h = hm._handle
        terms = torch.empty(b, 2, dtype=torch.float32, device=v.device)
        nsaved = L.tuch_hd_contact_saved_bytes(h, b)
        nws = L.tuch_hd_contact_workspace_bytes(h, b)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=v.device)
        ws = _workspace(nws, v.device)
        ext, md, part = exterior.contiguous(), _f32(min_d2), partner.contiguous()
        _C.check(L.tuch_hd_contact_fwd(h, _C.ptr(v), _C.ptr(ext), _C.ptr(md), _C.ptr(part), _C.ptr(valid), b, euclthres, thresh,
                                       _C.ptr(terms), _C.ptr(saved), nsaved, _C.ptr(ws), nws, _C.stream()))
        ctx.save_for_backward(saved)
        ctx.hm, ctx.shape = hm, tuple(v.shape)
        hm.last_saved = saved           # inspection only
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        saved, = ctx.saved_tensors
        hm = ctx.hm
        b = ctx.shape[0]
        L = _C.lib()
        g = grad_terms.to(torch.float32).contiguous()
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        nws = L.tuch_hd_contact_workspace_bytes(hm._handle, b)
        ws = _workspace(nws, g.device)
        _C.check(L.tuch_hd_contact_bwd(hm._handle, _C.ptr(saved), _C.ptr(g), b, _C.ptr(grad), _C.ptr(ws), nws, _C.stream()))
        return grad, None, None, None, None, None, None, None


class _RegionPairMin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, model: ContactModel, select, masked):
        v = _f32(verts)
        b = v.shape[0]
        out = torch.empty(b, model.num_pairs, dtype=torch.float32, device=v.device)
        ij = torch.empty(b, model.num_pairs, 2, dtype=torch.int32, device=v.device)
        sel = select.to(torch.uint8).contiguous() if select is not None else None
        _C.check(_C.lib().tuch_region_pair_min(model._handle, _C.ptr(v), b, _C.ptr(sel), int(masked),
                                               _C.ptr(out), _C.ptr(ij), _C.stream()))
        ctx.save_for_backward(v, ij)
        ctx.model = model
        ctx.mark_non_differentiable(ij)
        return out, ij

    @staticmethod
    def backward(ctx, grad_out, _grad_ij):
        v, ij = ctx.saved_tensors
        grad = torch.zeros_like(v)
        g_out = grad_out.to(torch.float32).contiguous()
        _C.check(_C.lib().tuch_region_pair_min_bwd(ctx.model._handle, _C.ptr(v), v.shape[0], _C.ptr(ij),
                                                   _C.ptr(g_out), _C.ptr(grad), _C.stream()))
        return grad, None, None, None

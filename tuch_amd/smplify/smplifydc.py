"""Drop-in for the reference's ``tuch/smplify/smplifydc.py``: the SMPLify-DC optimiser.

Same constructor and ``__call__`` signature / 7-tuple return as the reference
(smplifydc.py:27-276).  Differences, all behind the interface:
  * the assets the reference loads from disk inside ``__init__`` (SMPL .pkl, GMM prior,
    constants.JOINT_IDS) may be injected (``smpl=``, ``pose_prior=``, ``ign_joints=``) because
    they do not ship; with the licensed files present the reference's paths are used;
  * the device follows the inputs (the reference hard-codes 'cuda');
  * the contact term of every iteration is one batched pass over the HIP kernels.
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from .losses import (body_fitting_loss, camera_fitting_loss, contact_fitting_loss, contact_model_for,
                     stage2_objective)
from .prior import MaxMixturePrior
from .. import ops
from ..optim import make_adam

log = logging.getLogger(__name__)

# smplifydc.py:46-47: joints ignored during the fit, by name; resolved through constants.JOINT_IDS when the
# data folder is importable, else through SPIN's published table (models/smpl.py) -> [1, 9, 12, 27, 28]
IGNORED_JOINT_NAMES = ['OP Neck', 'OP RHip', 'OP LHip', 'Right Hip', 'Left Hip']


def default_ignored_joints():
    from ..models.smpl import spin_joint_ids
    ids = spin_joint_ids()
    return [ids[n] for n in IGNORED_JOINT_NAMES]


class SMPLifyDC():
    """SMPLify with discrete self-contact: stage 1 fits camera translation (+ betas when contact is
    used), stage 2 fits pose and global orientation against the contact objective."""

    def __init__(self,
                 step_size=1e-2,
                 batch_size=66,
                 num_iters=100,
                 focal_length=5000,
                 geodistssmpl=None,
                 geothres=0.0,
                 euclthres=0.0,
                 device=torch.device('cuda'),
                 smpl=None, pose_prior=None, ign_joints=None,
                 smpl_model_dir=None, prior_folder=None, use_graph=True, record_history=False):
        from ..assets import config_path
        self.device = device
        self.focal_length = focal_length
        self.step_size = step_size
        self.ign_joints = list(default_ignored_joints() if ign_joints is None else ign_joints)
        self._ign_index = {}          # device -> index tensor (a Python list index is re-uploaded on every use)
        self.num_iters = num_iters
        if pose_prior is None:             # smplifydc.py:50-52
            pose_prior = MaxMixturePrior(prior_folder=prior_folder or config_path('PRIOR_FOLDER'),
                                         num_gaussians=8, dtype=torch.float32)
        self.pose_prior = pose_prior.to(device)
        if smpl is None:                   # smplifydc.py:54-56
            from ..models.smpl import SMPL
            smpl = SMPL(smpl_model_dir or config_path('SMPL_MODEL_DIR'), batch_size=batch_size,
                        create_transl=False)
        self.smpl = smpl.to(self.device)
        self.face_tensor = torch.tensor(self.smpl.faces.astype(np.int64), dtype=torch.long,
                                        device=self.device).unsqueeze_(0).repeat([batch_size, 1, 1])
        self.geodistssmpl = geodistssmpl
        self.geothres = geothres
        self.geomask = self.geodistssmpl > self.geothres            # smplifydc.py:65 (strict >)
        self.euclthres = euclthres
        # replay each optimisation loop as a hipGraph after three eager iterations (same arithmetic,
        # no per-kernel launch cost: at small batch the loop is launch-bound otherwise)
        self.use_graph = use_graph
        # measurement / test aid (not in the reference): with record_history the objective and the parameters
        # *before* every update are kept per stage in self.history = {'stage1': [...], 'stage2': [...]}
        self.record_history = record_history
        # read once: TUCH_GRAPH_STRICT=1 turns a failed capture into an error (the tests set it), TUCH_SMPLIFY_SESSIONS=0
        # makes every call capture its loops afresh
        self.graph_strict = os.environ.get('TUCH_GRAPH_STRICT', '0') == '1'
        self.keep_sessions = os.environ.get('TUCH_SMPLIFY_SESSIONS', '1') != '0'
        self.fused_adam = os.environ.get('TUCH_FUSED_ADAM', '1') != '0'      # 0: Adam as a launch of its own (A/B, tests)
        self.history = None
        self.graph_replayed = {}
        # captured loops are kept between calls (keyed by batch size and the constant arguments): a training step
        # with SMPLify-DC in the loop runs 10 + 10 iterations per call, far too few to pay for two captures each time
        self._sessions = {}

    def _optimise(self, params, iteration, num_iters, adam_kwargs, collect=None, stage=''):
        """Run ``num_iters`` Adam iterations of ``iteration()`` (which returns (loss, vertices))."""
        graph_ok = self.use_graph and params[0].is_cuda and num_iters > 4
        # torch.optim.Adam's update as one launch where it applies (tuch_amd/optim.py), else torch's own
        optimizer = make_adam(params, self.step_size, capturable=graph_ok, **adam_kwargs)
        static = {}
        history = self.history[stage] if self.record_history else None

        def one():
            if history is not None:
                static['params'] = [p.detach().clone() for p in params]
            loss, verts = iteration()
            optimizer.zero_grad(set_to_none=graph_ok)
            ops.backward_scalar(loss)
            optimizer.step()
            static['verts'] = verts
            static['loss'] = loss.detach()
            return verts

        def keep():
            if history is not None:
                history.append({'loss': static['loss'].clone(), 'params': [p.clone() for p in static['params']]})

        done = 0
        if graph_ok:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    verts = one()
                    keep()
                    if collect is not None:
                        collect.append(verts.detach().clone())
            torch.cuda.current_stream().wait_stream(side)
            done = 3
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    one()
            except Exception as exc:
                # a loop that cannot be captured still runs (eagerly), but never silently: TUCH_GRAPH_STRICT=1
                # (set by the tests) turns this into an error
                if self.graph_strict:
                    raise
                log.warning('SMPLifyDC: hipGraph capture of the %s loop failed (%r); finishing with eager launches',
                            stage, exc)
                torch.cuda.synchronize()
                graph = None
            if graph is not None:
                for _ in range(num_iters - done):
                    graph.replay()
                    keep()
                    if collect is not None:
                        collect.append(static['verts'].detach().clone())
                self.graph_replayed[stage] = num_iters - done
                return
        for _ in range(num_iters - done):
            verts = one()
            keep()
            if collect is not None:
                collect.append(verts)

    # ------------------------------------------------------------------ loops kept between calls
    class _Stage:
        """One Adam loop on static tensors: three eager iterations + capture the first time, replays afterwards."""

        def __init__(self, owner, name, params, iteration, adam_kwargs, fuse_backward=False):
            self.owner, self.name, self.params, self.iteration = owner, name, params, iteration
            # fuse_backward (the contact stage 2: parameters = the body model's two pose tensors, objective = one root node):
            # the body model's last backward kernel applies Adam's update itself, step() is then a no-op (optim.py)
            self.optimizer = make_adam(params, owner.step_size, capturable=True, fuse_backward=fuse_backward, **adam_kwargs)
            self.graph, self.verts, self.loss = None, None, None

        def _one(self):
            loss, verts = self.iteration()
            self.optimizer.zero_grad(set_to_none=True)
            ops.backward_scalar(loss)
            self.optimizer.step()
            self.verts, self.loss = verts, loss.detach()

        def run(self, num_iters, collect):
            for state in self.optimizer.state.values():            # a fresh optimiser per call (smplifydc.py:117,150)
                for v in state.values():
                    if torch.is_tensor(v):
                        v.zero_()
            if torch.cuda.is_current_stream_capturing():
                # an ENCLOSING capture is recording (TUCH.forward_train_step --run_smplify captured as one hipGraph: BASELINE
                # config 5): a child graph cannot be replayed into it -- the iterations are unrolled into the enclosing graph
                for _ in range(num_iters):
                    self._one()
                    if collect is not None:
                        collect.append(self.verts.detach().clone())
                self.owner.graph_replayed[self.name] = 0
                return
            done = 0
            if self.graph is None:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(min(3, num_iters)):
                        self._one()
                        if collect is not None:
                            collect.append(self.verts.detach().clone())
                torch.cuda.current_stream().wait_stream(side)
                done = min(3, num_iters)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    self._one()
                self.graph = graph
            for _ in range(num_iters - done):
                self.graph.replay()
                if collect is not None:
                    collect.append(self.verts.detach().clone())
            self.owner.graph_replayed[self.name] = num_iters - done

    def _session(self, batch, use_contact, contactlist, segments, contact_loss_weight, with_pairs, like):
        # everything a captured loop bakes in: the tables (the session keeps them alive, so their ids cannot be reused by
        # other objects) and the fitter's own settings -- changing one of those after a call starts a new session
        key = (batch, bool(use_contact), id(contactlist), id(segments), float(contact_loss_weight), bool(with_pairs),
               float(self.step_size), float(self.euclthres), float(self.focal_length), tuple(self.ign_joints),
               id(self.pose_prior), int(self.num_iters))
        sess = self._sessions.get(key)
        if sess is not None:
            return sess
        dev, f32 = like.device, torch.float32
        z = lambda *shape, dtype=f32: torch.zeros(*shape, dtype=dtype, device=dev)
        t = dict(body_pose=z(batch, 69), global_orient=z(batch, 3), betas=z(batch, 10), cam=z(batch, 3), init_cam=z(batch, 3),
                 centre=z(batch, 2), j2d=z(batch, 49, 2), conf1=z(batch, 49), conf2=z(batch, 49),
                 valid=z(batch, dtype=torch.uint8))
        model = None
        if use_contact:
            model = contact_model_for(self.geomask, self.face_tensor, segments, contactlist, device=dev)
            t['select'] = z(batch, model.num_pairs, dtype=torch.uint8) if (with_pairs and model.num_pairs > 0) else None
        body_pose, global_orient, betas, cam = t['body_pose'], t['global_orient'], t['betas'], t['cam']
        spw = 1.0 if use_contact else 0.0

        def camera_iteration():
            out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas)
            return camera_fitting_loss(out, cam, t['init_cam'], t['centre'], t['j2d'], t['conf1'],
                                       focal_length=self.focal_length, shape_prior_weight=spw), out.vertices

        def contact_iteration():
            out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas)
            loss = stage2_objective(model, t['valid'], t['select'], body_pose, betas, out.joints, self.euclthres, cam,
                                    t['centre'], t['j2d'], t['conf2'], self.pose_prior, out.vertices,
                                    focal_length=self.focal_length, contact_loss_weight=contact_loss_weight,
                                    apply_segments=segments is not None)
            return loss, out.vertices

        def body_iteration():
            out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas)
            return body_fitting_loss(body_pose, betas, out.joints, cam, t['centre'], t['j2d'], t['conf2'], self.pose_prior,
                                     focal_length=self.focal_length), out.vertices

        def flags(bp, go, be, ct):
            body_pose.requires_grad, global_orient.requires_grad, betas.requires_grad, cam.requires_grad = bp, go, be, ct
        sess = dict(t=t, flags=flags, use_contact=use_contact, keys_alive=(contactlist, segments, self.pose_prior))
        # stage 1 optimises [betas, cam] with contact, [global_orient, cam] without (smplifydc.py:104-117)
        flags(False, not use_contact, bool(use_contact), True)
        sess['stage1'] = self._Stage(self, 'stage1', [betas, cam] if use_contact else [global_orient, cam], camera_iteration,
                                     dict(betas=(0.9, 0.999)))
        if use_contact:
            sess['stage2'] = (lambda: flags(True, True, False, False),
                              lambda: self._Stage(self, 'stage2', [body_pose, global_orient], contact_iteration, {},
                                                  fuse_backward=self.fused_adam))
        else:
            sess['stage2'] = (lambda: flags(True, True, True, False),
                              lambda: self._Stage(self, 'stage2', [body_pose, betas, global_orient], body_iteration,
                                                  dict(betas=(0.9, 0.999))))
        while len(self._sessions) >= 4:
            self._sessions.pop(next(iter(self._sessions)))
        self._sessions[key] = sess
        return sess

    def _call_cached(self, init_pose, init_betas, init_cam_t, camera_center, keypoints_2d, use_contact, contactlist,
                     gt_contact, ignore_idxs, has_discrete_contact, has_gt_keypoints, contact_loss_weight, segments):
        batch = init_pose.shape[0]
        with_pairs = gt_contact is not None and gt_contact[0] is not None
        sess = self._session(batch, use_contact, contactlist, segments, contact_loss_weight, with_pairs, init_pose)
        t = sess['t']
        with torch.no_grad():
            put = lambda dst, src: dst.copy_(src)
            put(t['body_pose'], init_pose[:, 3:]); put(t['global_orient'], init_pose[:, :3]); put(t['betas'], init_betas)
            put(t['cam'], init_cam_t); put(t['init_cam'], init_cam_t); put(t['centre'], camera_center)
            put(t['j2d'], keypoints_2d[:, :, :2]); put(t['conf1'], keypoints_2d[:, :, -1])
            put(t['conf2'], keypoints_2d[:, :, -1])
            t['conf2'].index_fill_(1, self._ignored(t['conf2'].device), 0.0)                                         # smplifydc.py:153,198
            if use_contact:
                put(t['valid'], ~ignore_idxs)
                if t.get('select') is not None:
                    put(t['select'], (gt_contact[0] == 1) & has_discrete_contact.bool()[:, None] & (~ignore_idxs)[:, None])
        sess['flags'](False, not use_contact, bool(use_contact), True)
        sess['stage1'].run(self.num_iters, None)
        set_flags, make_stage = sess['stage2']
        set_flags()
        if not isinstance(sess.get('stage2_obj'), self._Stage):
            sess['stage2_obj'] = make_stage()
        optiverts = []
        sess['stage2_obj'].run(self.num_iters, optiverts)
        with torch.no_grad():
            out = self.smpl(global_orient=t['global_orient'], body_pose=t['body_pose'], betas=t['betas'], return_full_pose=True)
            conf = t['conf2'].clone()
            if has_gt_keypoints is not None:                                             # smplifydc.py:219-220, no host sync
                conf[:, :25] = torch.where(has_gt_keypoints.bool()[:, None], torch.zeros_like(conf[:, :25]), conf[:, :25])
            reprojection_loss = body_fitting_loss(t['body_pose'], t['betas'], out.joints, t['cam'], t['centre'], t['j2d'], conf,
                                                  self.pose_prior, focal_length=self.focal_length, output='reprojection')
        pose = torch.cat([t['global_orient'], t['body_pose']], dim=-1).detach().clone()
        return (out.vertices.detach(), out.joints.detach(), pose, t['betas'].detach().clone(), t['cam'].detach().clone(),
                reprojection_loss, optiverts if optiverts else None)

    def __call__(self, init_pose, init_betas, init_cam_t,
                 camera_center, keypoints_2d, use_contact=False,
                 contactlist=[], gt_contact=None,
                 ignore_idxs=None, has_discrete_contact=None,
                 has_gt_keypoints=None, contact_loss_weight=1,
                 contact_loss_return='sum', segments=None):
        """Fit a batch of bodies.  Returns (vertices, joints, pose, betas, camera_translation,
        reprojection_loss, optiverts) exactly like the reference (smplifydc.py:231-236)."""
        if not (self.use_graph and init_pose.is_cuda):
            return self._fit(init_pose, init_betas, init_cam_t, camera_center, keypoints_2d, use_contact, contactlist,
                             gt_contact, ignore_idxs, has_discrete_contact, has_gt_keypoints, contact_loss_weight,
                             contact_loss_return, segments)
        from ..ops import off_default_stream
        with off_default_stream(init_pose.device):       # graph replays never run on the NULL stream (ops.py)
            return self._fit(init_pose, init_betas, init_cam_t, camera_center, keypoints_2d, use_contact, contactlist,
                             gt_contact, ignore_idxs, has_discrete_contact, has_gt_keypoints, contact_loss_weight,
                             contact_loss_return, segments)

    def _fit(self, init_pose, init_betas, init_cam_t, camera_center, keypoints_2d, use_contact, contactlist, gt_contact,
             ignore_idxs, has_discrete_contact, has_gt_keypoints, contact_loss_weight, contact_loss_return, segments):
        if (self.use_graph and init_pose.is_cuda and self.num_iters > 4 and not self.record_history
                and self.keep_sessions
                and (not use_contact or (ignore_idxs is not None and isinstance(contactlist, (dict, list))))):
            try:
                return self._call_cached(init_pose, init_betas, init_cam_t, camera_center, keypoints_2d, use_contact,
                                         contactlist, gt_contact, ignore_idxs, has_discrete_contact, has_gt_keypoints,
                                         contact_loss_weight, segments)
            except Exception as exc:
                if self.graph_strict:
                    raise
                log.warning('SMPLifyDC: the cached hipGraph loops failed (%r); running this call without them', exc)
                self._sessions.clear()
                torch.cuda.synchronize()
        if self.record_history:
            self.history = {'stage1': [], 'stage2': []}
        camera_translation = init_cam_t.clone()
        joints_2d = keypoints_2d[:, :, :2].contiguous()
        joints_conf = keypoints_2d[:, :, -1].clone()
        body_pose = init_pose[:, 3:].detach().clone()
        global_orient = init_pose[:, :3].detach().clone()
        betas = init_betas.detach().clone()

        # ---- stage 1: camera translation (+ shape with contact, + orientation without)
        body_pose.requires_grad = False
        camera_translation.requires_grad = True
        global_orient.requires_grad = not use_contact
        betas.requires_grad = bool(use_contact)
        stage1 = [betas, camera_translation] if use_contact else [global_orient, camera_translation]
        shape_prior_weight = 1.0 if use_contact else 0.0

        def camera_iteration():
            out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas)
            return camera_fitting_loss(out, camera_translation, init_cam_t, camera_center, joints_2d,
                                       joints_conf, focal_length=self.focal_length,
                                       shape_prior_weight=shape_prior_weight), out.vertices

        self._optimise(stage1, camera_iteration, self.num_iters, dict(betas=(0.9, 0.999)), stage='stage1')

        # ---- stage 2: pose + global orientation
        optiverts = []
        joints_conf.index_fill_(1, self._ignored(joints_conf.device), 0.0)                        # smplifydc.py:153,198
        camera_translation.requires_grad = False
        # snapshots of the stage-1 result (smplifydc.py:141-142), taken before gradients are switched on:
        # a clone of a leaf that requires grad would keep its AccumulateGrad node (created on the
        # default stream) alive and break the graph capture of the loop below
        pose_stage1 = body_pose.detach().clone()
        orient_stage1 = global_orient.detach().clone()
        body_pose.requires_grad = True
        global_orient.requires_grad = True
        if use_contact:
            betas.requires_grad = False

            def contact_iteration():
                out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas)
                loss = contact_fitting_loss(body_pose, global_orient, pose_stage1, orient_stage1,
                                            betas, out.joints, self.geomask, self.euclthres,
                                            camera_translation, camera_center, joints_2d, joints_conf,
                                            self.pose_prior, cdict=contactlist, gt_contact=gt_contact,
                                            ignore_idxs=ignore_idxs,
                                            has_discrete_contact=has_discrete_contact,
                                            verts=out.vertices, face_tensor=self.face_tensor,
                                            focal_length=self.focal_length,
                                            contact_loss_weight=contact_loss_weight,
                                            output=contact_loss_return, segments=segments)
                return loss, out.vertices

            self._optimise([body_pose, global_orient], contact_iteration, self.num_iters, {}, optiverts, stage='stage2')
        else:
            betas.requires_grad = True

            def body_iteration():
                out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas)
                return body_fitting_loss(body_pose, betas, out.joints, camera_translation, camera_center,
                                         joints_2d, joints_conf, self.pose_prior,
                                         focal_length=self.focal_length), out.vertices

            self._optimise([body_pose, betas, global_orient], body_iteration, self.num_iters,
                           dict(betas=(0.9, 0.999)), optiverts, stage='stage2')
        if len(optiverts) == 0:
            optiverts = None

        # ---- final evaluation
        with torch.no_grad():
            out = self.smpl(global_orient=global_orient, body_pose=body_pose, betas=betas,
                            return_full_pose=True)
            if has_gt_keypoints is not None:
                joints_conf[has_gt_keypoints, :25] = 0
            reprojection_loss = body_fitting_loss(body_pose, betas, out.joints, camera_translation,
                                                  camera_center, joints_2d, joints_conf, self.pose_prior,
                                                  focal_length=self.focal_length, output='reprojection')
        pose = torch.cat([global_orient, body_pose], dim=-1).detach()
        return (out.vertices.detach(), out.joints.detach(), pose, betas.detach(), camera_translation,
                reprojection_loss, optiverts)

    def _ignored(self, device):
        """ign_joints as an index tensor on `device`."""
        idx = self._ign_index.get(device)
        if idx is None:
            idx = self._ign_index[device] = torch.tensor(self.ign_joints, dtype=torch.int64, device=device)
        return idx

    def get_fitting_loss(self, pose, betas, cam_t, camera_center, keypoints_2d, has_gt_keypoints=None):
        """Per-joint reprojection loss of given parameters (reference: smplifydc.py:238-276;
        like the reference, zeroing the ignored joints writes through into ``keypoints_2d``)."""
        joints_2d = keypoints_2d[:, :, :2]
        joints_conf = keypoints_2d[:, :, -1]
        joints_conf.index_fill_(1, self._ignored(joints_conf.device), 0.0)
        if has_gt_keypoints is not None:
            joints_conf = joints_conf.clone()
            joints_conf[has_gt_keypoints, :25] = 0
        with torch.no_grad():
            out = self.smpl(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas,
                            return_full_pose=True)
            return body_fitting_loss(pose[:, 3:], betas, out.joints, cam_t, camera_center, joints_2d,
                                     joints_conf, self.pose_prior, focal_length=self.focal_length,
                                     output='reprojection')

"""CPU: the compat shim maps the reference's module paths onto tuch_amd and keeps the
reference's call signatures (argument names and order) on the hot-path functions."""
import inspect
import sys


def test_install_maps_reference_module_paths():
    import tuch_amd.compat as compat
    saved = {k: v for k, v in sys.modules.items() if k == 'tuch' or k.startswith('tuch.')}
    try:
        names = compat.install()
        assert 'tuch.smplify.losses' in names
        from tuch.smplify.losses import contact_fitting_loss
        from tuch.smplify.smplifydc import SMPLifyDC
        from tuch.utils.contact import batch_pairwise_dist, solid_angles, winding_numbers
        import tuch_amd.smplify.losses as ours
        assert contact_fitting_loss is ours.contact_fitting_loss
        assert callable(SMPLifyDC) and callable(batch_pairwise_dist) and callable(solid_angles)
        assert callable(winding_numbers)
    finally:
        for k in [k for k in sys.modules if k == 'tuch' or k.startswith('tuch.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_signatures_match_the_reference():
    from tuch_amd.smplify import losses, smplifydc
    from tuch_amd.train.loss import RegressorLoss
    from tuch_amd.utils import contact
    # tuch/smplify/losses.py:34-48
    want = ['body_pose', 'global_orient', 'body_pose_loop1', 'opt_global_orient_smplifyloop1', 'betas',
            'model_joints', 'geomask', 'euclthres', 'camera_t', 'camera_center', 'joints_2d', 'joints_conf',
            'pose_prior', 'cdict', 'gt_contact', 'ignore_idxs', 'has_discrete_contact', 'verts', 'face_tensor',
            'device', 'focal_length', 'sigma', 'pose_prior_weight', 'shape_prior_weight', 'angle_prior_weight',
            'contact_loss_weight', 'output', 'segments']
    assert list(inspect.signature(losses.contact_fitting_loss).parameters) == want
    # tuch/smplify/smplifydc.py:68-73
    call = list(inspect.signature(smplifydc.SMPLifyDC.__call__).parameters)[1:]
    assert call == ['init_pose', 'init_betas', 'init_cam_t', 'camera_center', 'keypoints_2d', 'use_contact',
                    'contactlist', 'gt_contact', 'ignore_idxs', 'has_discrete_contact', 'has_gt_keypoints',
                    'contact_loss_weight', 'contact_loss_return', 'segments']
    init = list(inspect.signature(smplifydc.SMPLifyDC.__init__).parameters)[1:9]
    assert init == ['step_size', 'batch_size', 'num_iters', 'focal_length', 'geodistssmpl', 'geothres',
                    'euclthres', 'device']
    # tuch/train/loss.py:45-56, 94-111
    assert list(inspect.signature(RegressorLoss.__init__).parameters)[1:10] == [
        'options', 'device', 'num_verts', 'faces', 'geodistssmpl', 'geothres', 'euclthres', 'face_tensor', 'use_hd']
    assert list(inspect.signature(RegressorLoss.forward).parameters)[1:] == [
        'pred_rotmat', 'pred_betas', 'opt_pose', 'opt_betas', 'pred_keypoints_2d', 'gt_keypoints_2d',
        'pred_joints', 'gt_joints', 'has_pose_3d', 'pred_vertices', 'opt_vertices', 'pred_camera', 'valid_fit',
        'valid_fit_shape']
    # tuch/utils/contact.py:23,49,112
    assert list(inspect.signature(contact.batch_pairwise_dist).parameters) == ['x', 'y', 'use_cuda', 'squared']
    assert list(inspect.signature(contact.winding_numbers).parameters) == ['points', 'triangles', 'thresh']


def test_product_never_imports_the_oracle():
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tuch_amd')
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f

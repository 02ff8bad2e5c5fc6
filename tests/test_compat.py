"""CPU: the compat shim maps the reference's module paths onto tuch_amd and keeps the
reference's call signatures (argument names and order) on the hot-path functions."""
import inspect
import os
import sys
import types

import pytest


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


# ---- unmapped tuch.* modules keep resolving to the reference checkout (train.py:27-33, demo_smplify_dc.py:29-36) ----

def _purge(prefixes):
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in prefixes}
    for k in saved:
        del sys.modules[k]
    return saved


def _fake_checkout(root):
    """A stand-in for the reference checkout, written by the test: three unmapped modules and one file
    under a *mapped* name that must never be the one that is imported."""
    import os
    files = {
        'tuch/utils/saver.py': "WHO = 'checkout saver'\n",
        'tuch/models/hmr.py': "from tuch.utils.geometry import rot6d_to_rotmat\nWHO = 'checkout hmr'\n",
        'tuch/train/trainer.py': "from tuch.utils.saver import WHO as SAVER\nfrom .fits_dict import FitsDict\n"
                                 "WHO = 'checkout trainer'\n",
        'tuch/datasets/base_dataset.py': "WHO = 'checkout dataset'\n",
        'tuch/smplify/smplifydc.py': "raise ImportError('the checkout file under a mapped name was imported')\n",
        'tuch/train/fits_dict.py': "raise ImportError('the checkout file under a mapped name was imported')\n",
    }
    for rel, text in files.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(text)


@pytest.mark.parametrize('path_first', [True, False])
def test_unmapped_modules_come_from_the_checkout(tmp_path, path_first):
    import importlib
    import tuch_amd.compat as compat
    import tuch_amd.smplify.smplifydc as ours
    import tuch_amd.train.fits_dict as our_fits
    _fake_checkout(str(tmp_path))
    saved = _purge(('tuch', 'torchgeometry'))
    try:
        if path_first:
            sys.path.insert(0, str(tmp_path))
            compat.install()
        else:                                   # the checkout joins sys.path after install()
            compat.install()
            sys.path.append(str(tmp_path))
        importlib.invalidate_caches()
        import tuch.utils.saver
        import tuch.models.hmr
        import tuch.train.trainer
        from tuch.datasets.base_dataset import WHO
        assert tuch.utils.saver.WHO == 'checkout saver' and WHO == 'checkout dataset'
        assert tuch.models.hmr.WHO == 'checkout hmr' and tuch.train.trainer.WHO == 'checkout trainer'
        assert tuch.models.hmr.__file__.startswith(str(tmp_path))
        # mapped names: ours, whatever the checkout holds under the same name
        import tuch.smplify.smplifydc
        assert tuch.smplify.smplifydc is ours and tuch.train.trainer.FitsDict is our_fits.FitsDict
        import tuch_amd.utils.geometry as our_geometry
        assert tuch.models.hmr.rot6d_to_rotmat is our_geometry.rot6d_to_rotmat
        # ... also after somebody dropped the entry from sys.modules
        del sys.modules['tuch.smplify.smplifydc']
        assert importlib.import_module('tuch.smplify.smplifydc') is ours
    finally:
        compat.uninstall()
        if str(tmp_path) in sys.path:
            sys.path.remove(str(tmp_path))
        sys.modules.update(saved)
        importlib.invalidate_caches()


class _Anything(types.ModuleType):
    """Stand-in for a third-party package that is absent from this image (torchvision, cv2, pyrender, ...)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        sub = _Anything(self.__name__ + '.' + name)
        sys.modules[sub.__name__] = sub
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return self

    def __mro_entries__(self, bases):
        return (object,)


class _StubFinder:
    """Serves `stub.sub.module` for every package stubbed by the test below."""
    roots = ()

    @classmethod
    def find_spec(cls, fullname, path=None, target=None):
        import importlib.machinery
        if fullname.split('.')[0] not in cls.roots and not any(fullname.startswith(r + '.') for r in cls.roots):
            return None

        class _L:
            @staticmethod
            def create_module(spec):
                return _Anything(spec.name)

            @staticmethod
            def exec_module(module):
                pass
        return importlib.machinery.ModuleSpec(fullname, _L, is_package=True)


_THIRD_PARTY = ('torchvision', 'cv2', 'skimage', 'pyrender', 'trimesh', 'matplotlib', 'tensorboard', 'smplx',
                'scipy.misc', 'joblib', 'tqdm', 'torch.utils.tensorboard', 'OpenGL', 'tensorboardX')
REFERENCE = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='the reference checkout exists in the build container only')
@pytest.mark.parametrize('script,first,last', [('train.py', 24, 34), ('demo_smplify_dc.py', 24, 37)])
def test_reference_script_import_blocks_run_after_install(tmp_path, script, first, last):
    """The import block of the reference's own script, executed as written, with stubs for third-party packages
    that are absent here and for nothing else; data/ is the synthetic tree in the reference's formats."""
    import importlib
    import tuch_amd.compat as compat
    from synthetic import make_body, write_reference_assets
    write_reference_assets(make_body(10, 12), str(tmp_path))
    saved = _purge(('tuch', 'configs', 'data', 'torchgeometry') + tuple(t.split('.')[0] for t in _THIRD_PARTY
                                                                         if t.split('.')[0] != 'torch'))
    stubbed = []
    cwd = os.getcwd()
    try:
        for name in _THIRD_PARTY:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Anything(name)
                stubbed.append(name)
        _StubFinder.roots = tuple(stubbed)
        sys.meta_path.append(_StubFinder)
        os.chdir(str(tmp_path))
        sys.path[:0] = [str(tmp_path), REFERENCE]
        importlib.invalidate_caches()
        compat.install()
        lines = open(os.path.join(REFERENCE, script)).read().split('\n')[first - 1:last]
        block = '\n'.join(l for l in lines if l.startswith(('import ', 'from ')))
        assert 'tuch.models.hmr' in block and 'tuch.smplify.smplifydc' in block
        scope = {}
        exec(compile(block, script, 'exec'), scope)
        import tuch_amd.smplify.smplifydc as ours
        import tuch_amd.models.smpl as our_smpl
        assert scope['SMPLifyDC'] is ours.SMPLifyDC and scope['SMPL'] is our_smpl.SMPL
        assert scope['hmr'].__module__ == 'tuch.models.hmr'
        assert sys.modules['tuch.models.hmr'].__file__.startswith(REFERENCE)
        if script == 'train.py':
            import tuch_amd.train.loss as our_loss
            import tuch_amd.train.train_module as our_tm
            assert scope['RegressorLoss'] is our_loss.RegressorLoss and scope['TUCH'] is our_tm.TUCH
            assert sys.modules['tuch.train.trainer'].__file__.startswith(REFERENCE)
            assert sys.modules['tuch.datasets.mixed_dataset'].__file__.startswith(REFERENCE)
        else:
            assert sys.modules['tuch.utils.renderer'].__file__.startswith(REFERENCE)
            assert sys.modules['tuch.datasets.base_dataset'].__file__.startswith(REFERENCE)
    finally:
        os.chdir(cwd)
        compat.uninstall()
        if _StubFinder in sys.meta_path:
            sys.meta_path.remove(_StubFinder)
        for p in (str(tmp_path), REFERENCE):
            if p in sys.path:
                sys.path.remove(p)
        for k in [k for k in sys.modules if k.split('.')[0] in ('configs', 'data')] + stubbed:
            sys.modules.pop(k, None)
        for k in [k for k in sys.modules if any(k.startswith(s + '.') for s in stubbed)]:
            del sys.modules[k]
        sys.modules.update(saved)
        importlib.invalidate_caches()

"""BASELINE config 1: the reference's CPU plumbing through ``EgoNet`` (no GPU).

``get_keypoints(is_cuda=False)`` / ``lift_2d_to_3d(cuda=False)`` / ``get_6d_rep`` /
``post_process`` on a CPU model, as the reference's fixtures were made
(libs/model/egonet.py:424-486, 279-295).  HC and L run torch on the CPU (the reference's own
PyTorch-CPU path); the crop->screen affine and the pose solve run the host twins of the device
kernels (egn_keypoints_to_screen_host_f64, egn_pose_solve_host_f64: same pose_math.h).  Checked
against the outputs of the REFERENCE (tests/golden/egonet_pipeline.npz, egonet_w48_pipeline.npz).
"""
import json

import numpy as np
import torch

from conftest import golden, fixture_cfg, sd_crc, arr_crc, require_same_rng
from egonet_amd import configs, synth
from egonet_amd.model.egonet import EgoNet


def _ego(cfg, g, hc_seed, l_seed):
    ego = EgoNet(cfg, pre_trained=False)
    hc_sd = synth.synth_state_dict(ego.HC.state_dict(), seed=hc_seed)
    require_same_rng(sd_crc(hc_sd), g['hc_crc'], 'HC weights')
    ego.HC.load_state_dict(hc_sd)
    ego.L.load_state_dict(synth.synth_state_dict(ego.L.state_dict(), seed=l_seed))
    ego.LS = {k[3:]: g[k] for k in g.files if k.startswith('ls/')}
    return ego.eval()


def test_egonet_cpu_route_vs_reference_tiny():
    g = golden('egonet_pipeline.npz')
    ego = _ego(fixture_cfg(g), g, 6, 7)
    crops = synth.synth_crops(6, 3, 64, 64, seed=8)
    boxes = g['boxes']
    records = ego.make_records({'path': ['img0.png', 'img1.png'], 'boxes': [boxes[:3], boxes[3:]]})
    rec = ego.get_keypoints(crops, records, is_cuda=False)
    rec = ego.lift_2d_to_3d(rec, cuda=False)
    kp2d = np.concatenate([np.concatenate(rec[p]['kpts_2d_pred']) for p in rec])
    kp3d = np.concatenate([rec[p]['kpts_3d_pred'] for p in rec])
    np.testing.assert_allclose(kp2d, g['kpts_2d'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(kp3d, g['kpts_3d'], rtol=0, atol=1e-4)
    for p in rec:
        rec[p]['K'] = g['K']
    for mode, key in (('proj', 'alpha_proj'), ('trans', 'alpha_trans')):
        out = ego.post_process({p: dict(r) for p, r in rec.items()}, alpha_mode=mode)
        al = np.concatenate([out[p]['alphas'] for p in out])
        eu = np.concatenate([out[p]['euler_angles'] for p in out])
        np.testing.assert_allclose((1 + np.cos(al - g[key])) / 2, 1.0, rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.cos(eu), np.cos(g['euler']), atol=1e-5)
        np.testing.assert_allclose(np.concatenate([out[p]['translation'] for p in out]), g['translation'], atol=1e-4)
    # pose angles on the reference's own 3D points: the host solve itself, to 1e-8
    e, t = ego.get_6d_rep(g['kpts_3d'])
    np.testing.assert_allclose(np.cos(e), np.cos(g['euler']), atol=1e-8)
    np.testing.assert_allclose(np.sin(e), np.sin(g['euler']), atol=1e-8)


def test_config1_single_crop_w48_demo_topology_on_cpu(tmp_path):
    """One 256x256 crop through the demo.yml topology (HRNet-W48, coordinates head, 1024-wide lifter)
    on PyTorch-CPU, through the reference-shaped API, against the reference's run of the same crop."""
    g = golden('egonet_w48_pipeline.npz')
    ego = _ego(configs.w48_config('coordinates'), g, 1, 2)
    crops = synth.synth_crops(16, 3, 256, 256, seed=12)
    require_same_rng(arr_crc(crops.numpy()), g['crops_crc'], 'crops')
    annot = {'path': ['frame00.png'], 'boxes': [g['boxes'][:1]], 'scores': [g['scores'][:1]]}
    records = ego.make_records(annot)
    np.testing.assert_allclose(records[0]['center'], g['centers'][0], atol=1e-12)
    rec = ego.get_keypoints(crops[:1], records, is_cuda=False)
    rec = ego.lift_2d_to_3d(rec, cuda=False)
    r = rec['frame00.png']
    np.testing.assert_allclose(r['kpts_2d_pred'][0], g['kpts_2d'][:1], rtol=0, atol=1e-3)
    np.testing.assert_allclose(r['kpts_3d_pred'], g['kpts_3d'][:1], rtol=0, atol=1e-3)
    r['K'] = g['K']
    r['raw_txt_format'] = json.loads(str(g['raw_txt']))['frame00.png'][:1]
    out = ego.post_process(rec, alpha_mode='proj', save_dict={'flag': True, 'save_dir': str(tmp_path)})
    al = out['frame00.png']['alphas']
    np.testing.assert_allclose((1 + np.cos(al - g['alpha_proj'][:1])) / 2, 1.0, rtol=0, atol=1e-6)
    want = json.loads(str(g['pred_str']))['frame00.png'].strip().split('\n')[0].split()
    got = open(str(tmp_path / 'frame00.txt')).read().split()
    assert len(got) == len(want) == 16
    for i, (a, b) in enumerate(zip(got, want)):
        if i in (3, 14):
            assert abs(float(a) - float(b)) < 5e-4
        else:
            assert a == b


def test_get_keypoints_with_rotated_records_vs_reference():
    """[round 6] libs/model/egonet.py:442-452: every record's 'rotation' goes into get_affine_transform(inv=1).  The
    fixture holds the REFERENCE's screen coordinates for seven records with rotations 0 / 30 / -45 / 90 / 12.5 / 180 /
    -7.25 degrees (tests/golden/make_golden.py section 9); HC is replaced by a stub that returns the fixture's local
    coordinates, so this checks the crop -> screen step alone (rot = 0 rows take the host twin of the device kernel)."""
    g = golden('kpts_rotated.npz')
    ego = EgoNet(configs.tiny_config('coordinates'), pre_trained=False).eval()
    ego.resolution = [int(v) for v in g['resolution']]
    local = torch.from_numpy(g['local'])
    n = local.shape[0]

    class _Stub(torch.nn.Module):
        def forward(self, x):
            return None, local
    ego.HC = _Stub()
    records = [dict(path='img%d.png' % (i // 4), center=g['centers'][i], scale=g['scales'][i], rotation=float(g['rots'][i]),
                    bbox_resize=[0, 0, 1, 1], label='Car', score=1.0) for i in range(n)]
    rec = ego.get_keypoints(torch.zeros(n, 3, 8, 8), records, is_cuda=False)
    got = np.concatenate([np.concatenate(rec[p]['kpts_2d_pred']) for p in rec]).reshape(n, -1, 2)
    np.testing.assert_allclose(got, g['screen'], rtol=0, atol=1e-6)
    assert [r for p in rec for r in rec[p]['rotation']] == [float(v) for v in g['rots']]

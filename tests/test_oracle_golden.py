"""Pin the CPU oracle (oracle/) against outputs of the REFERENCE itself.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py,
which imports /root/reference in the build container.  Weights and inputs are
regenerated here from egonet_amd.synth (seeded per state_dict key); their CRCs
are stored in the fixtures.
"""
import numpy as np
import pytest
import torch

from conftest import golden, fixture_cfg, sd_crc, arr_crc, require_same_rng
from egonet_amd import configs, synth
from egonet_amd.model.heatmapModel import hrnet as hip_hrnet
from egonet_amd.model import FCmodel as hip_fc
from oracle import hrnet_oracle, decode_oracle, lifter_oracle, geometry_oracle


def _synth_hc(cfg, seed):
    net = hip_hrnet.get_pose_net(cfg, is_train=False)
    return synth.synth_state_dict(net.state_dict(), seed=seed)


@pytest.mark.parametrize('name', ['tiny_coords', 'tiny_heatmap', 'tiny_ped'])
def test_hrnet_oracle_tiny(name):
    g = golden('hrnet_%s.npz' % name)
    cfg = fixture_cfg(g)
    sd = _synth_hc(cfg, seed=3)
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    iw, ih = cfg['heatmapModel']['input_size']
    x = synth.synth_crops(int(g['n']), 3, ih, iw, seed=5)
    require_same_rng(arr_crc(x.numpy()), g['x_crc'], 'input')
    out = hrnet_oracle.hrnet_forward(sd, cfg, x)
    if isinstance(out, tuple):
        np.testing.assert_allclose(out[0].numpy(), g['maps'], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out[1].numpy(), g['coords'], rtol=0, atol=1e-6)
    else:
        np.testing.assert_allclose(out.numpy(), g['maps'], rtol=0, atol=1e-5)


@pytest.mark.parametrize('head', ['coordinates', 'heatmap'])
def test_hrnet_oracle_w48(head):
    """Full HRNet-W48 @256x256 on 4 crops: heat-map samples, arg-max indices
    (bit exact), coordinates, and both decodes."""
    g = golden('hrnet_w48_outputs.npz')
    cfg = configs.w48_config(head)
    sd = _synth_hc(cfg, seed=1)
    assert len(sd) == int(g[head + '/n_keys'])
    require_same_rng(sd_crc(sd), g[head + '/sd_crc'], 'weights')
    x = synth.synth_crops(4, 3, 256, 256, seed=11)
    require_same_rng(arr_crc(x.numpy()), g[head + '/x_crc'], 'input')
    torch.set_num_threads(max(torch.get_num_threads(), 4))
    out = hrnet_oracle.hrnet_forward(sd, cfg, x)
    maps = (out[0] if isinstance(out, tuple) else out).numpy()
    np.testing.assert_allclose(maps[:, :, ::4, ::4], g[head + '/maps_sub'], rtol=0, atol=2e-4)
    idx, mv = decode_oracle.argmax_index(maps)
    assert np.array_equal(idx, g[head + '/argmax'])
    np.testing.assert_allclose(mv[..., 0], g[head + '/maxval'], rtol=0, atol=2e-4)
    if isinstance(out, tuple):
        np.testing.assert_allclose(out[1].numpy(), g[head + '/coords'], rtol=0, atol=1e-5)
    hard, _ = decode_oracle.get_max_preds(maps)
    assert np.array_equal(hard, g[head + '/hard_preds'])
    soft, smv = decode_oracle.soft_arg_max(maps)
    np.testing.assert_allclose(soft, g[head + '/soft_preds'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(smv, g[head + '/soft_maxvals'], rtol=0, atol=2e-4)


def test_decode_oracle_edge_cases():
    g = golden('decode.npz')
    hm = g['hm']
    hard, mv = decode_oracle.get_max_preds(hm)
    assert np.array_equal(hard, g['hard_preds'])           # masked map, tie, all-equal map
    assert np.array_equal(mv, g['hard_maxvals'])
    soft, smv = decode_oracle.soft_arg_max(hm)
    np.testing.assert_allclose(soft, g['soft_preds'], rtol=0, atol=1e-4)
    assert np.array_equal(smv, g['soft_maxvals'])
    p, pm = decode_oracle.soft_arg_max_np(g['pos'])
    np.testing.assert_allclose(p, g['np_preds'], rtol=0, atol=1e-4)
    assert np.array_equal(pm, g['np_maxvals'])


def test_lifter_oracle_full():
    g = golden('lifter_full.npz')
    cfg = configs.w48_config()
    net = hip_fc.get_fc_model(1, cfg, 66, 96)
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    assert len(sd) == int(g['n_keys']) == 37
    assert sum(p.numel() for p in net.parameters()) == int(g['n_params']) == 4375648
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    y = lifter_oracle.lifter_forward(sd, torch.from_numpy(g['x']))
    np.testing.assert_allclose(y.numpy(), g['y'], rtol=0, atol=1e-5)


@pytest.mark.parametrize('leaky', [False, True])
def test_lifter_oracle_tiny_stored_weights(leaky):
    g = golden('lifter_tiny%s.npz' % ('_leaky' if leaky else ''))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    y = lifter_oracle.lifter_forward(sd, torch.from_numpy(g['x']), leaky=leaky)
    np.testing.assert_allclose(y.numpy(), g['y'], rtol=0, atol=1e-5)


def test_pipeline_oracle():
    """EgoNet.get_keypoints -> lift_2d_to_3d -> get_6d_rep -> alpha of the
    reference (CPU run, tiny HC, 33 joints)."""
    g = golden('egonet_pipeline.npz')
    cfg = fixture_cfg(g)
    hc_sd = _synth_hc(cfg, seed=6)
    lnet = hip_fc.get_fc_model(1, cfg, 66, 96)
    l_sd = synth.synth_state_dict(lnet.state_dict(), seed=7)
    require_same_rng(sd_crc(hc_sd), g['hc_crc'], 'HC weights')
    require_same_rng(sd_crc(l_sd), g['l_crc'], 'L weights')
    crops = synth.synth_crops(6, 3, 64, 64, seed=8)
    require_same_rng(arr_crc(crops.numpy()), g['crops_crc'], 'crops')
    boxes = synth.synth_boxes(6, seed=2)
    np.testing.assert_array_equal(boxes, g['boxes'])
    stats = {k[3:]: g[k] for k in g.files if k.startswith('ls/')}
    _, coords = hrnet_oracle.hrnet_forward(hc_sd, cfg, crops)
    local = decode_oracle.coords_head_to_pixels(coords.numpy(), cfg['heatmapModel']['input_size'])
    kp2d = []
    for i, b in enumerate(boxes):
        ret = geometry_oracle.modify_bbox(b, 1.0)
        np.testing.assert_allclose(ret['c'], g['centers'][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(ret['s'], g['scales'][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(ret['bbox'], g['bbox_resize'][i], rtol=0, atol=1e-10)
        kp2d.append(geometry_oracle.crop_to_screen(local[i], ret['c'], ret['s'], (64, 64)).reshape(1, -1))
    kp2d = np.concatenate(kp2d)
    np.testing.assert_allclose(kp2d, g['kpts_2d'], rtol=0, atol=1e-3)
    kp3d = lifter_oracle.lift_2d_to_3d(l_sd, stats, kp2d)
    np.testing.assert_allclose(kp3d, g['kpts_3d'], rtol=0, atol=1e-3)
    # pose solve on the reference's own 3D predictions
    euler, trans = geometry_oracle.six_dof(g['kpts_3d'])
    np.testing.assert_allclose(np.cos(euler), np.cos(g['euler']), atol=1e-9)
    np.testing.assert_allclose(np.sin(euler), np.sin(g['euler']), atol=1e-9)
    np.testing.assert_allclose(trans, g['translation'], atol=1e-12)
    a = geometry_oracle.observation_angle_proj(g['euler'], g['kpts_2d'][:, 0], g['K'])
    np.testing.assert_allclose(a, g['alpha_proj'], atol=1e-12)
    a = geometry_oracle.observation_angle_trans(g['euler'], g['translation'])
    np.testing.assert_allclose(a, g['alpha_trans'], atol=1e-12)


def test_pose_oracle_cuboids():
    g = golden('pose_solve.npz')
    euler, trans = geometry_oracle.six_dof(g['preds'])
    np.testing.assert_allclose(euler, g['euler'], atol=1e-9)
    np.testing.assert_allclose(trans, g['translation'], atol=0)
    np.testing.assert_allclose(geometry_oracle.observation_angle_proj(euler, g['kpts_x'], g['K']),
                               g['alpha_proj'], atol=1e-9)
    np.testing.assert_allclose(geometry_oracle.observation_angle_trans(euler, trans),
                               g['alpha_trans'], atol=1e-9)


@pytest.mark.parametrize('tag', ['s1', 's2', 'rect'])
def test_target_oracle_vs_reference_generate_target(tag):
    from oracle import targets_oracle
    g = golden('targets.npz')
    tgt, w = targets_oracle.generate_target_batch(g[tag + '/joints'], g[tag + '/vis'], g[tag + '/input_size'],
                                                  g[tag + '/heatmap_size'], int(g[tag + '/sigma']))
    assert tgt.dtype == np.float32 and tgt.shape == g[tag + '/target'].shape
    np.testing.assert_array_equal(tgt, g[tag + '/target'])          # same float32 arithmetic: bit exact
    np.testing.assert_array_equal(w, g[tag + '/weight'])
    if tag == 's1':                              # the fixture holds the edge cases
        assert g[tag + '/weight'][0, 2, 0] == 0 and g[tag + '/vis'][0, 2] == 1      # dot entirely outside
        assert 0 < g[tag + '/target'][0, 3].max() < 1                                # clipped dot, centre outside


def test_hrnet_train_oracle_vs_reference():
    """Two train-mode iterations of the reference HRNet (tiny topology) with the
    reference's JointsCompositeLoss and Adam: losses, first-step outputs and
    gradients, parameters and running statistics after the second step."""
    import json
    from oracle.hrnet_train_oracle import HRNetTrainOracle
    g = golden('hrnet_train.npz')
    cfg = fixture_cfg(g)
    sd = _synth_hc(cfg, seed=21)
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    keys = json.loads(str(g['keys']))
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3)
    assert orc.param_keys == json.loads(str(g['param_order']))
    losses = []
    for it in range(2):
        x = synth.synth_crops(4, 3, 64, 64, seed=30 + it)
        loss, maps, coords = orc.step(x, torch.from_numpy(g['target'][it]), torch.from_numpy(g['joints'][it][..., :2]))
        losses.append(loss)
        if it == 0:
            np.testing.assert_allclose(maps.numpy(), g['maps1'], rtol=0, atol=1e-5)
            np.testing.assert_allclose(coords.numpy(), g['coords1'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(losses, g['losses'], rtol=1e-6)
    fin = orc.sd
    for k in keys:
        # Adam's first steps move every entry by ~lr * sign(grad): an entry whose
        # gradient is rounding noise may step the other way
        d = np.abs(fin[k].detach().numpy() - g['p2/' + k])
        assert np.mean(d > 1e-5) < 0.01, k
    for k in ('bn1.running_mean', 'bn1.running_var', 'stage3.0.branches.2.0.bn1.running_var',
              'head2.1.bn2.running_mean'):
        np.testing.assert_allclose(fin[k].numpy(), g['p2/' + k], rtol=1e-4, atol=1e-5, err_msg=k)
    assert int(fin['bn1.num_batches_tracked']) == int(g['p2/bn1.num_batches_tracked']) == 2


def test_hrnet_train_oracle_first_step_gradients():
    import json
    from oracle.hrnet_train_oracle import HRNetTrainOracle
    g = golden('hrnet_train.npz')
    cfg = fixture_cfg(g)
    orc = HRNetTrainOracle(_synth_hc(cfg, seed=21), cfg, lr=1e-3)
    x = synth.synth_crops(4, 3, 64, 64, seed=30)
    orc.step(x, torch.from_numpy(g['target'][0]), torch.from_numpy(g['joints'][0][..., :2]), update=False)
    grads = orc.grads()
    norms = np.array([float(grads[k].double().norm()) for k in orc.param_keys])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=1e-4, atol=1e-9)
    for k in json.loads(str(g['keys'])):
        ref = g['g1/' + k]
        np.testing.assert_allclose(grads[k].numpy(), ref, rtol=0, atol=1e-5 * max(1e-6, float(np.abs(ref).max())),
                                   err_msg=k)


def test_lifter_train_oracle_vs_reference():
    """Three train-mode iterations (batch-stat BN, MSE(mean), Adam) of the
    reference's FCModel on CPU."""
    from oracle.lifter_train_oracle import LifterTrainOracle
    g = golden('lifter_train.npz')
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd0/')}
    sd3 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd3/')}
    orc = LifterTrainOracle(sd0, lr=1e-3)
    losses = [orc.step(torch.from_numpy(g['xs'][i]), torch.from_numpy(g['ys'][i])) for i in range(3)]
    np.testing.assert_allclose(losses, g['losses'], rtol=1e-6)
    for k, v in sd3.items():
        np.testing.assert_allclose(orc.sd[k].detach().numpy(), v.numpy(), rtol=0, atol=2e-6, err_msg=k)
    assert int(orc.sd['batch_norm1.num_batches_tracked']) == 3


CR_CASES = [(t, s, th) for t in ('rand', 'wide', 'cluster') for s in ('sl1', 'l1', 'mse') for th in (0.15, 0.1)]


@pytest.mark.parametrize('tag,spec,thres', CR_CASES)
def test_cross_ratio_oracle_vs_reference(tag, spec, thres):
    """Value, line mask and gradient of the reference's cross-ratio term
    (function.py:113-153) on the 'bbox12' lines."""
    from oracle.hrnet_train_oracle import cross_ratio_loss, cross_ratio_mask
    from egonet_amd.common.img_proc import get_cr_indices
    g = golden('cr_loss.npz')
    idx = g['cr_indices']
    np.testing.assert_array_equal(get_cr_indices(), idx)
    key = '%s/%s/%g' % (tag, spec, thres)
    c = torch.from_numpy(g[tag + '/coords']).clone().requires_grad_(True)
    np.testing.assert_array_equal(cross_ratio_mask(c, idx, thres).numpy(), g[key + '/mask'][..., 0])
    loss = cross_ratio_loss(c, idx, 4 / 3, thres, spec)
    loss.backward()
    ref = g[key + '/grad']
    np.testing.assert_allclose(float(loss), float(g[key + '/loss']), rtol=2e-6, atol=0)
    np.testing.assert_allclose(c.grad.numpy(), ref, rtol=0, atol=2e-6 * max(1.0, float(np.abs(ref).max())))
    if tag == 'cluster':
        assert g[key + '/mask'].sum() == 0 and float(loss) == 0.0      # every line fore-shortened
    else:
        assert 0 < g[key + '/mask'].sum() < g[key + '/mask'].size


def test_composite_loss_with_cross_ratio_vs_reference():
    from oracle.hrnet_train_oracle import composite_loss
    g = golden('cr_loss.npz')
    maps = torch.from_numpy(g['full/maps']).clone().requires_grad_(True)
    c = torch.from_numpy(g['rand/coords']).clone().requires_grad_(True)
    loss = composite_loss((maps, c), torch.from_numpy(g['full/target']), torch.from_numpy(g['full/joints'][..., :2]),
                          (256, 256), 1.0, 0.1, w_cr=0.05, cr_indices=g['cr_indices'])
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g['full/loss']), rtol=2e-6)
    np.testing.assert_allclose(c.grad.numpy(), g['full/dcoords'], rtol=0, atol=1e-6 * float(np.abs(g['full/dcoords']).max()))
    np.testing.assert_allclose(maps.grad.numpy(), g['full/dmaps'], rtol=0, atol=1e-9)


def test_joints_mse_loss_oracle_vs_reference():
    """oracle.hrnet_train_oracle.joints_mse_loss restates libs/loss/function.py:22-46 (JointsMSELoss, with and without
    use_target_weight): value and gradient equal the reference's own class on tests/golden/jmse_loss.npz."""
    import torch
    from oracle import hrnet_train_oracle as hto
    g = golden('jmse_loss.npz')
    tgt, tw = torch.from_numpy(g['target']), torch.from_numpy(g['target_weight'])
    for flag in (0, 1):
        p = torch.from_numpy(g['pred']).clone().requires_grad_(True)
        loss = hto.joints_mse_loss(p, tgt, tw if flag else None)
        loss.backward()
        assert abs(float(loss.detach()) - float(g['loss_%d' % flag])) < 1e-6 * abs(float(g['loss_%d' % flag]))
        np.testing.assert_allclose(p.grad.numpy(), g['grad_%d' % flag], rtol=0, atol=1e-8)
    assert float(g['loss_0']) != float(g['loss_1'])

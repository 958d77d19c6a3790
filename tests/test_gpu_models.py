"""End-to-end parity on a real MI355X through the reference-shaped operator API
(get_pose_net / get_fc_model / EgoNet) -> C ABI -> HIP kernels, against
 (a) the committed golden outputs of the REFERENCE (tests/golden/), and
 (b) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): key-point coordinates and lifted 3D points
within 1e-3 abs (fp32), heat-map arg-max indices bit exact.
"""
import numpy as np
import pytest
import torch

from conftest import golden, fixture_cfg, sd_crc, arr_crc, require_same_rng
from egonet_amd import configs, synth
from egonet_amd.model.heatmapModel import hrnet as hip_hrnet
from egonet_amd.model import FCmodel as hip_fc
from oracle import hrnet_oracle, decode_oracle, lifter_oracle, geometry_oracle

pytestmark = pytest.mark.gpu


def _model(cfg, seed):
    net = hip_hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=seed)
    net.load_state_dict(sd)
    return net.eval().cuda(), sd


@pytest.mark.parametrize('name', ['tiny_coords', 'tiny_heatmap', 'tiny_ped'])
def test_hrnet_tiny_vs_reference_outputs(name):
    g = golden('hrnet_%s.npz' % name)
    cfg = fixture_cfg(g)
    net, sd = _model(cfg, 3)
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    iw, ih = cfg['heatmapModel']['input_size']
    x = synth.synth_crops(int(g['n']), 3, ih, iw, seed=5)
    with torch.no_grad():
        out = net(x.cuda())
    torch.cuda.synchronize()
    maps = (out[0] if isinstance(out, tuple) else out).cpu().numpy()
    np.testing.assert_allclose(maps, g['maps'], rtol=0, atol=2e-4)
    if isinstance(out, tuple):
        assert out[1].shape == g['coords'].shape
        np.testing.assert_allclose(out[1].cpu().numpy(), g['coords'], rtol=0, atol=2e-5)
    idx_ref = g['maps'].reshape(maps.shape[0], maps.shape[1], -1).argmax(axis=2)
    assert np.array_equal(maps.reshape(maps.shape[0], maps.shape[1], -1).argmax(axis=2), idx_ref)


@pytest.mark.parametrize('head', ['coordinates', 'heatmap'])
def test_hrnet_w48_vs_reference_outputs(head):
    """The headline model: HRNet-W48 @256x256, 4 crops, both heads + decode."""
    from egonet_amd.common import img_proc
    g = golden('hrnet_w48_outputs.npz')
    cfg = configs.w48_config(head)
    net, sd = _model(cfg, 1)
    require_same_rng(sd_crc(sd), g[head + '/sd_crc'], 'weights')
    x = synth.synth_crops(4, 3, 256, 256, seed=11)
    with torch.no_grad():
        out = net(x.cuda())
    maps_d = out[0] if isinstance(out, tuple) else out
    maps = maps_d.cpu().numpy()
    assert maps.shape == (4, 33, 64, 64)
    np.testing.assert_allclose(maps[:, :, ::4, ::4], g[head + '/maps_sub'], rtol=0, atol=5e-4)
    # bit-exact arg-max indices through the decode kernel
    xy, mx, idx = img_proc.hard_arg_max(maps_d)
    assert np.array_equal(idx.cpu().numpy(), g[head + '/argmax'])
    assert np.array_equal(xy.cpu().numpy(), g[head + '/hard_preds'])
    np.testing.assert_allclose(mx.cpu().numpy()[..., 0], g[head + '/maxval'], rtol=0, atol=5e-4)
    sxy, _ = img_proc.soft_arg_max(maps_d)
    np.testing.assert_allclose(sxy.cpu().numpy(), g[head + '/soft_preds'], rtol=0, atol=1e-3)
    if isinstance(out, tuple):
        # coordinates in crop pixels within 1e-3
        np.testing.assert_allclose(out[1].cpu().numpy() * 256, g[head + '/coords'] * 256, rtol=0, atol=1e-3)
    # fused decode inside the program gives the same answer
    (o2, dec) = net._hip_engine().forward(x.cuda(), decode_mode=1)
    np.testing.assert_allclose(dec[0].cpu().numpy(), sxy.cpu().numpy(), rtol=0, atol=1e-6)


def test_hrnet_batch_sizes_and_determinism():
    cfg = configs.tiny_config('heatmap')
    net, sd = _model(cfg, 9)
    for n in (1, 3, 8):
        x = synth.synth_crops(n, 3, 64, 64, seed=n)
        with torch.no_grad():
            a = net(x.cuda()).cpu()
            b = net(x.cuda()).cpu()
        assert torch.equal(a, b)
        want = hrnet_oracle.hrnet_forward(sd, cfg, x)
        np.testing.assert_allclose(a.numpy(), want.numpy(), rtol=0, atol=2e-4)
    # weights changed -> engine must repack
    sd2 = synth.synth_state_dict(net.state_dict(), seed=10)
    net.load_state_dict(sd2)
    x = synth.synth_crops(2, 3, 64, 64, seed=1)
    with torch.no_grad():
        a = net(x.cuda()).cpu()
    want = hrnet_oracle.hrnet_forward({k: v.cpu() for k, v in sd2.items()}, cfg, x)
    np.testing.assert_allclose(a.numpy(), want.numpy(), rtol=0, atol=2e-4)
    with pytest.raises(ValueError), torch.no_grad():
        net(torch.zeros(1, 3, 48, 64, device='cuda'))


def test_lifter_vs_reference_outputs():
    g = golden('lifter_full.npz')
    cfg = configs.w48_config()
    net = hip_fc.get_fc_model(1, cfg, 66, 96)
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    net.load_state_dict(sd)
    net = net.eval().cuda()
    with torch.no_grad():
        y = net(torch.from_numpy(g['x']).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g['y'], rtol=0, atol=1e-3)
    # ragged batches incl. 1 and empty
    for n in (1, 5, 100):
        x = torch.randn(n, 66, generator=torch.Generator().manual_seed(n))
        with torch.no_grad():
            y = net(x.cuda()).cpu()
        want = lifter_oracle.lifter_forward(sd, x)
        np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=0, atol=1e-3)
    with torch.no_grad():
        assert net(torch.zeros(0, 66, device='cuda')).shape == (0, 96)


@pytest.mark.parametrize('leaky', [False, True])
def test_lifter_tiny_stored_weights(leaky):
    g = golden('lifter_tiny%s.npz' % ('_leaky' if leaky else ''))
    c2 = configs.tiny_config()
    c2['FCModel']['leaky'] = leaky
    net = hip_fc.get_fc_model(1, c2, 10, 12)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')})
    net = net.eval().cuda()
    with torch.no_grad():
        y = net(torch.from_numpy(g['x']).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g['y'], rtol=0, atol=1e-4)


def test_egonet_pipeline_vs_reference_outputs():
    """EgoNet.get_keypoints / lift_2d_to_3d / get_6d_rep / alpha + the batched
    infer_crops path, against the reference's CPU run (tiny HC, 33 joints)."""
    from egonet_amd.model.egonet import EgoNet
    g = golden('egonet_pipeline.npz')
    cfg = fixture_cfg(g)
    ego = EgoNet(cfg, pre_trained=False)
    hc_sd = synth.synth_state_dict(ego.HC.state_dict(), seed=6)
    l_sd = synth.synth_state_dict(ego.L.state_dict(), seed=7)
    require_same_rng(sd_crc(hc_sd), g['hc_crc'], 'HC weights')
    ego.HC.load_state_dict(hc_sd)
    ego.L.load_state_dict(l_sd)
    ego.LS = {k[3:]: g[k] for k in g.files if k.startswith('ls/')}
    ego = ego.eval().cuda()
    crops = synth.synth_crops(6, 3, 64, 64, seed=8)
    boxes = g['boxes']
    annot = {'path': ['img0.png', 'img1.png'], 'boxes': [boxes[:3], boxes[3:]]}
    records = ego.make_records(annot)
    np.testing.assert_allclose(np.stack([r['center'] for r in records]), g['centers'], atol=1e-12)
    rec = ego.get_keypoints(crops, records)
    rec = ego.lift_2d_to_3d(rec)
    kp2d = np.concatenate([np.concatenate(rec[p]['kpts_2d_pred']) for p in rec])
    kp3d = np.concatenate([rec[p]['kpts_3d_pred'] for p in rec])
    np.testing.assert_allclose(kp2d, g['kpts_2d'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(kp3d, g['kpts_3d'], rtol=0, atol=1e-3)
    for p in rec:
        rec[p]['K'] = g['K']
    rec = ego.post_process(rec, alpha_mode='proj')
    # pose angles on the REFERENCE's 3D points (conditioning-independent check)
    e, t = ego.get_6d_rep(g['kpts_3d'])
    np.testing.assert_allclose(np.cos(e), np.cos(g['euler']), atol=1e-8)
    np.testing.assert_allclose(np.sin(e), np.sin(g['euler']), atol=1e-8)
    # batched device pipeline == record API
    res = ego.infer_crops(crops.cuda(), g['centers'], g['scales'], K=g['K'])
    # (the crop centres come from two float64 computations that agree to 1e-13; the
    # reference rounds them to float32 control points, which may flip one ulp)
    np.testing.assert_allclose(res['kpts_2d'], kp2d, rtol=0, atol=1e-4)
    np.testing.assert_allclose(res['kpts_3d'], kp3d, rtol=0, atol=1e-5)
    al = np.concatenate([rec[p]['alphas'] for p in rec])
    np.testing.assert_allclose(np.cos(res['alpha']), np.cos(al), atol=1e-6)


def test_program_timing_and_graph_replay():
    cfg = configs.tiny_config('heatmap')
    net, sd = _model(cfg, 4)
    x = synth.synth_crops(2, 3, 64, 64, seed=2).cuda()
    eng = net._hip_engine()
    with torch.no_grad():
        ref = net(x).cpu()
    eng.forward(x, timed=True)
    ms = eng.last_ms
    prog = eng.program(x)
    assert len(ms) == len(prog.meta) and (ms >= 0).all() and ms.sum() > 0
    # hipGraph capture + replay of the same launch list
    out = torch.empty_like(ref, device='cuda')
    prog.bind(2, x)
    prog.bind(3, out)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        prog.capture()
        out.zero_()
        prog.replay()
    s.synchronize()
    assert torch.equal(out.cpu(), ref)

"""End-to-end parity on a real MI355X through the reference-shaped operator API
(get_pose_net / get_fc_model / EgoNet) -> C ABI -> HIP kernels, against
 (a) the committed golden outputs of the REFERENCE (tests/golden/), and
 (b) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): key-point coordinates and lifted 3D points
within 1e-3 abs (fp32), heat-map arg-max indices bit exact.
"""
import os

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


def _kind_counts(net):
    """{filter kind: launches} and {cfg: launches} over the conv launches of the programs the engine has built."""
    from egonet_amd import _lib
    L = _lib.lib()
    kinds, cfgs = {}, {}
    for prog in net._hip_engine().programs.values():
        for m in prog.meta:
            if m['kind'] == 'conv':
                k = L.egn_conv_config_kind(m['cfg']) if m['cfg'] > 0 else 0
                kinds[k] = kinds.get(k, 0) + 1
                cfgs[m['cfg']] = cfgs.get(m['cfg'], 0) + 1
    return kinds, cfgs


@pytest.mark.parametrize('wino', ['43', '43b', '1', '0'])
@pytest.mark.parametrize('head', ['coordinates', 'heatmap'])
def test_hrnet_w48_vs_reference_outputs(head, wino, monkeypatch):
    """The headline model: HRNet-W48 @256x256, 4 crops, both heads + decode, against the REFERENCE's own outputs --
    with the fused Winograd F(4x4,3x3) kernels forced onto every 3x3 s1 layer they plan for (EGONET_AMD_WINO=43:
    conv_wino4_kernel on the 64 x 64 / 32 x 32 maps + conv_wino4b_kernel on the 16 x 16 maps; =43b:
    conv_wino4b_kernel on all three; both: conv_wino4c_kernel on the 8 x 8 maps -- the bench's default kernels,
    VERDICT r3 weak #1), with the F(2x2,3x3) kernels
    (=1, csrc/conv_wino.hip), and with the direct kernels only (=0): every kernel family meets the reference's bar,
    arg-max indices and hard predictions bit-exact."""
    from egonet_amd.common import img_proc
    monkeypatch.setenv('EGONET_AMD_WINO', wino)
    g = golden('hrnet_w48_outputs.npz')
    cfg = configs.w48_config(head)
    net, sd = _model(cfg, 1)
    require_same_rng(sd_crc(sd), g[head + '/sd_crc'], 'weights')
    x = synth.synth_crops(4, 3, 256, 256, seed=11)
    with torch.no_grad():
        out = net(x.cuda())
    kinds, cfgs = _kind_counts(net)
    if wino in ('43', '43b'):
        # 3x3 s1 layers of the 48 / 96 / 192 / 384-channel branches (+ the 256 -> 48 transition): F(4x4,3x3) -- the
        # 384-channel 8 x 8 maps on conv_wino4c_kernel with the K split (cfg 83, one launch per layer: ticket words);
        # the 64-channel layers: F(2x2,3x3)
        assert kinds.get(3, 0) >= 204 and kinds.get(1, 0) >= 4 and cfgs.get(83, 0) == 24, (kinds, cfgs)
        if wino == '43':
            assert cfgs.get(70, 0) >= 100 and cfgs.get(80, 0) >= 50, cfgs
        else:
            assert cfgs.get(80, 0) >= 180 and cfgs.get(70, 0) == 0, cfgs
    elif wino == '1':
        assert kinds.get(1, 0) >= 200 and 3 not in kinds and 2 not in kinds, kinds
    else:
        assert set(kinds) == {0}, kinds
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


@pytest.mark.parametrize('tag', ['tiny_pixshuf', 'tiny_angle'])
def test_hrnet_other_heads_vs_reference_outputs(tag):
    """Pixel-shuffle upsampler behind the heat-map head and the 'angleregression' head (reference
    hrnet.py:373-422, 598-611) as HIP programs: final_layer -> 1x1 conv + BN + ReLU -> PixelShuffle
    fused with the NCHW hand-over; 1x1 conv -> 4 strided BasicBlocks -> (AvgPool2d(4) + Linear + BN1d +
    ReLU as one 4x4 valid conv) -> Linear.  Against the reference's outputs, launch counter checked."""
    from egonet_amd import _lib
    g = golden('hrnet_%s.npz' % tag)
    cfg = fixture_cfg(g)
    net, sd = _model(cfg, 3)
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    iw, ih = cfg['heatmapModel']['input_size']
    x = synth.synth_crops(int(g['n']), 3, ih, iw, seed=5).cuda()
    c0 = _lib.lib().egn_launch_count()
    out = net(x)                                            # eval mode, grad on: the HIP program
    assert _lib.lib().egn_launch_count() - c0 > 40
    assert tuple(out.shape) == g['out'].shape
    np.testing.assert_allclose(out.cpu().numpy(), g['out'], rtol=0, atol=2e-4)
    if tag == 'tiny_pixshuf':
        idx_ref = g['out'].reshape(out.shape[0], out.shape[1], -1).argmax(axis=2)
        assert np.array_equal(out.cpu().numpy().reshape(out.shape[0], out.shape[1], -1).argmax(axis=2), idx_ref)


@pytest.mark.parametrize('mode', ['table', 'f43'])
def test_egonet_w48_pipeline_vs_reference_outputs(tmp_path, mode, monkeypatch):
    """BASELINE config 5 at full size on one GPU: HRNet-W48 (coordinates head) -> x256 -> crop affine
    -> lifter -> pose solve -> KITTI result lines for 16 crops of 4 frames, against the REFERENCE's
    own CPU run (tests/golden/egonet_w48_pipeline.npz, tools/inference.py:135-199 minus file I/O).
    get_keypoints is called the way the reference calls it -- eval mode, autograd enabled
    (libs/model/egonet.py:434) -- and must run the HIP program (launch counter).
    mode 'table': the SHIPPED tile table, autotuning off -- configs[4]'s per-GPU shard runs a reproducible
    selection with no shape left to the cost model; mode 'f43': the F(4x4,3x3) kernels forced
    onto every layer they plan for (>= 180 launches of filter kind 3)."""
    from egonet_amd import _lib
    from egonet_amd.model.egonet import EgoNet
    import json
    monkeypatch.setenv('EGONET_AMD_AUTOTUNE', '0')
    if mode == 'f43':
        monkeypatch.setenv('EGONET_AMD_WINO', '43')
    else:
        monkeypatch.delenv('EGONET_AMD_WINO', raising=False)
    g = golden('egonet_w48_pipeline.npz')
    cfg = configs.w48_config('coordinates')
    ego = EgoNet(cfg, pre_trained=False)
    hc_sd = synth.synth_state_dict(ego.HC.state_dict(), seed=1)
    l_sd = synth.synth_state_dict(ego.L.state_dict(), seed=2)
    require_same_rng(sd_crc(hc_sd), g['hc_crc'], 'HC weights')
    ego.HC.load_state_dict(hc_sd)
    ego.L.load_state_dict(l_sd)
    ego.LS = {k[3:]: g[k] for k in g.files if k.startswith('ls/')}
    ego = ego.eval().cuda()
    crops = synth.synth_crops(16, 3, 256, 256, seed=12)
    require_same_rng(arr_crc(crops.numpy()), g['crops_crc'], 'crops')
    boxes, scores = g['boxes'], g['scores']
    paths = ['frame%02d.png' % i for i in range(4)]
    annot = {'path': paths, 'boxes': [boxes[4 * i:4 * i + 4] for i in range(4)],
             'scores': [scores[4 * i:4 * i + 4] for i in range(4)]}
    records = ego.make_records(annot)
    L = _lib.lib()
    assert torch.is_grad_enabled()
    before = L.egn_launch_count()
    rec = ego.get_keypoints(crops, records)                 # no torch.no_grad() around it
    nops = sum(1 for m in ego.HC._hip_engine().program(crops.cuda()).meta if m['kind'] not in ('fork', 'join'))
    assert L.egn_launch_count() - before >= nops > 300
    kinds, cfgs = _kind_counts(ego.HC)
    if mode == 'f43':
        assert kinds.get(3, 0) >= 180 and cfgs.get(70, 0) >= 100, (kinds, cfgs)
    else:
        # every conv shape of the 16-crop program has a MEASURED entry in tuned/gfx950.json: nothing falls back to
        # the cost model (cfg 0), nothing is autotuned on the box
        assert cfgs.get(0, 0) == 0, 'shapes left to the cost model: %d launches' % cfgs.get(0, 0)
    rec = ego.lift_2d_to_3d(rec)
    kp2d = np.concatenate([np.concatenate(rec[p]['kpts_2d_pred']) for p in paths])
    kp3d = np.concatenate([rec[p]['kpts_3d_pred'] for p in paths])
    np.testing.assert_allclose(kp2d, g['kpts_2d'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(kp3d, g['kpts_3d'], rtol=0, atol=1e-3)
    raws = json.loads(str(g['raw_txt']))
    for p in paths:
        rec[p]['K'] = g['K']
        rec[p]['raw_txt_format'] = raws[p]
    rec = ego.post_process(rec, alpha_mode='proj', save_dict={'flag': True, 'save_dir': str(tmp_path)})
    al = np.concatenate([rec[p]['alphas'] for p in paths])
    eu = np.concatenate([rec[p]['euler_angles'] for p in paths])
    # AOS sees a prediction only through (1 + cos(delta alpha)) / 2 (evaluate_object_3d_offline.cpp:547-548)
    np.testing.assert_allclose((1 + np.cos(al - g['alpha_proj'])) / 2, 1.0, rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.cos(eu), np.cos(g['euler']), atol=2e-5)
    np.testing.assert_allclose(np.sin(eu), np.sin(g['euler']), atol=2e-5)
    # result lines (and the files post_process wrote): every field the reference copies from the
    # detection is byte-equal; alpha and rotation_y are printed with 6 decimals and follow the fp32
    # network output, so they agree to the key-point tolerance (format parity with the reference's
    # own angles is byte-exact: tests/test_format_cpu.py)
    want = json.loads(str(g['pred_str']))
    for p in paths:
        got_lines, want_lines = rec[p]['pred_str'].strip().split('\n'), want[p].strip().split('\n')
        assert len(got_lines) == len(want_lines) == 4
        with open(str(tmp_path / (p[:-4] + '.txt'))) as f:
            assert f.read() == rec[p]['pred_str']
        for la, lb in zip(got_lines, want_lines):
            a, b = la.split(), lb.split()
            assert len(a) == len(b) == 16
            for i, (ta, tb) in enumerate(zip(a, b)):
                if i in (3, 14):                            # alpha, rotation_y
                    d = abs(float(ta) - float(tb))
                    assert min(d, abs(d - 2 * np.pi)) < 5e-4, (i, ta, tb)
                else:
                    assert ta == tb, (i, ta, tb)
    # the batched device pipeline gives the same numbers
    res = ego.infer_crops(crops.cuda(), g['centers'], g['scales'], K=g['K'])
    np.testing.assert_allclose(res['kpts_2d'], g['kpts_2d'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(res['kpts_3d'], g['kpts_3d'], rtol=0, atol=1e-3)
    np.testing.assert_allclose((1 + np.cos(res['alpha'] - g['alpha_proj'])) / 2, 1.0, rtol=0, atol=1e-6)


def test_eval_mode_cuda_forward_always_runs_the_hip_program():
    """There is no torch/MIOpen route for an eval-mode CUDA tensor: with autograd ENABLED (the
    reference's validation loop, libs/trainer/trainer.py:421) HC and L still run the native program --
    the library's launch counter advances by the program's kernel count -- and return plain
    tensors.  CPU tensors run torch on the CPU (the reference's CPU path)."""
    from egonet_amd import _lib
    L = _lib.lib()
    cfg = configs.tiny_config('coordinates')
    net, sd = _model(cfg, 3)
    x = synth.synth_crops(2, 3, 64, 64, seed=5).cuda()
    with torch.no_grad():
        want = [t.clone() for t in net(x)]
    nops = sum(1 for m in net._hip_engine().program(x).meta if m['kind'] not in ('fork', 'join'))
    assert torch.is_grad_enabled() and not net.training
    c0 = L.egn_launch_count()
    got = net(x)                                            # grad mode on
    assert L.egn_launch_count() - c0 == nops
    assert all(not t.requires_grad and t.grad_fn is None for t in got)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    lif = hip_fc.get_fc_model(1, cfg, 10, 12)
    lif.load_state_dict(synth.synth_state_dict(lif.state_dict(), seed=4))
    lif = lif.eval().cuda()
    lif(torch.randn(9, 10).cuda())          # (shapes outside the shipped table are tuned on first use: programs of their own)
    c0 = L.egn_launch_count()
    y = lif(torch.randn(9, 10).cuda())                      # grad mode on
    assert L.egn_launch_count() - c0 == 6 + 1 and not y.requires_grad      # input relayout + six fused GEMMs
    c0 = L.egn_launch_count()
    net.cpu()(x.cpu())
    assert L.egn_launch_count() == c0                       # CPU tensors never touch the library


def test_engine_notices_every_weight_and_buffer_change():
    """The packed blob folds ALL parameters and BatchNorm buffers.  A change that leaves the FIRST
    parameter alone -- torch-autograd fine-tuning with freeze_layers: ['conv1', ...]
    (KITTI_train_IGRs_Ped.yml), a BatchNorm running-stat update -- must still rebuild it, and
    .train() drops the programs."""
    cfg = configs.tiny_config('coordinates')
    net, sd = _model(cfg, 3)
    x = synth.synth_crops(2, 3, 64, 64, seed=5).cuda()
    y0 = net(x)[0].clone()
    eng = net._engine
    assert eng is not None and len(eng.programs) == 1
    with torch.no_grad():                                   # what an optimizer step / BN update does
        net.stage3[0].branches[1][0].conv1.weight.mul_(1.5)
        net.stage2[0].branches[0][0].bn1.running_mean.add_(0.3)
    y1 = net(x)[0].clone()
    assert float((y1 - y0).abs().max()) > 1e-4
    fresh, _ = _model(cfg, 3)
    fresh.load_state_dict(net.state_dict())
    np.testing.assert_array_equal(fresh(x)[0].cpu().numpy(), y1.cpu().numpy())
    net.train()
    assert net._engine is None                              # dropped on .train()
    net.eval()
    np.testing.assert_array_equal(net(x)[0].cpu().numpy(), y1.cpu().numpy())
    # ... an nn.DataParallel replica (shallow copy with its own parameter tensors, re-created on every
    # forward) folds ITS parameters, not the engine it inherited from the original module
    rep = torch.nn.parallel.replicate(net, [0, 0])[1]
    assert rep._engine is net._engine                       # what the shallow copy starts with
    np.testing.assert_array_equal(rep(x)[0].cpu().numpy(), y1.cpu().numpy())
    assert rep._engine is not net._engine and rep._engine.model is rep
    # a replica with different weights (its tensors are copies) computes with ITS weights
    rep2 = torch.nn.parallel.replicate(net, [0, 0])[1]
    with torch.no_grad():
        rep2.stage2[0].branches[0][0].conv1.weight.mul_(0.5)
    assert float((rep2(x)[0] - y1).abs().max()) > 1e-4
    np.testing.assert_array_equal(net(x)[0].cpu().numpy(), y1.cpu().numpy())      # the original is untouched
    # nn.DataParallel itself (one device: the module is called directly)
    dp = torch.nn.DataParallel(net, device_ids=[0])
    np.testing.assert_array_equal(dp(x)[0].cpu().numpy(), y1.cpu().numpy())


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


@pytest.mark.parametrize('head', ['heatmap', 'coordinates'])
def test_small_batches_replay_their_program_as_a_hipgraph(head):
    """[round 5] engine.HRNetEngine._forward_graphed (opt-in: EGONET_AMD_GRAPH_MAX_N=<n>): batches of up to n crops run
    their program eagerly twice, then replay it as ONE hipGraph with static input / output tensors.  The case itself is
    tests/graph_case.py; it runs in a process of its own: inside the whole suite ``hipGraphLaunch`` crashed the
    interpreter (host segmentation fault in the runtime, reproducibly after tests/test_gpu_autograd.py had run in the
    same process, never alone or after any other module: profiles/r5_graph_replay_crash.txt) -- which is also why the
    feature is off by default."""
    import subprocess
    import sys
    env = dict(os.environ, EGONET_AMD_GRAPH_MAX_N='16')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'graph_case.py'), head],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and 'graph case ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize('head', ['heatmap', 'coordinates'])
def test_oversized_batches_are_chunked(head, monkeypatch):
    """[round 6] engine.HRNetEngine._forward_chunked: the reference takes any loader batch (libs/trainer/trainer.py:113-125,
    tools/inference.py:135-199); the kernels address tensors with 32-bit byte offsets, so a batch whose widest tensor
    would reach the limit is cut into the largest table-covered chunk sizes and the results are concatenated -- not
    refused.  Tiny model with the limit lowered to what 5 crops need: 13 crops run as 4 + 4 + 4 + 1, bit-identical to
    running those chunks one by one, through forward (+ decode) and through the module."""
    cfg = configs.tiny_config(head)
    net, sd = _model(cfg, 6)
    x = synth.synth_crops(13, 3, 64, 64, seed=41).cuda()
    eng = net._hip_engine()
    per_crop = 4 * max(256 * 16 * 16, 64 * 32 * 32, 3 * 64 * 64)
    monkeypatch.setenv('EGONET_AMD_MAX_TENSOR_BYTES', str(5 * per_crop))
    assert eng.max_batch(3, 64, 64) == 5
    got = eng.forward(x, decode_mode=1)
    parts = [eng.forward(x[a:b], decode_mode=1) for a, b in ((0, 4), (4, 8), (8, 12), (12, 13))]
    flat_g = torch.utils._pytree.tree_leaves(got)
    for k, g in enumerate(flat_g):
        want = torch.cat([torch.utils._pytree.tree_leaves(p)[k] for p in parts])
        assert g.shape[0] == 13 and torch.equal(g, want)
    with torch.no_grad():
        y = net(x)
    for k, g in enumerate(torch.utils._pytree.tree_leaves(y)):
        assert torch.equal(g, flat_g[k])
    monkeypatch.delenv('EGONET_AMD_MAX_TENSOR_BYTES')
    assert eng.max_batch(3, 256, 256) == 511
    whole = eng.forward(x, decode_mode=1)            # 13 crops fit one program: same values up to the kernels' batch-size choice
    for a, b in zip(torch.utils._pytree.tree_leaves(whole), flat_g):
        assert a.shape == b.shape and (a.float() - b.float()).abs().max().item() < 1e-3


def test_w48_batch_of_520_crops_runs_in_chunks():
    """The real limit: 520 crops of 256 x 256 (layer1's 256-channel tensor would be 2.03 GiB) = 4 x 128 + 8, equal bit for
    bit to the chunks run one by one; 511 crops is the largest single program."""
    cfg = configs.w48_config('heatmap')
    net, sd = _model(cfg, 9)
    eng = net._hip_engine()
    assert eng.max_batch(3, 256, 256) == 511
    base = synth.synth_crops(8, 3, 256, 256, seed=43).cuda()
    x = base.repeat(65, 1, 1, 1)
    x += torch.arange(520, device='cuda', dtype=torch.float32).view(-1, 1, 1, 1) * 1e-3
    got, (xy, mx, idx) = eng.forward(x, decode_mode=1)
    assert got.shape[0] == 520 and xy.shape[0] == 520
    at = 0
    for c in (128, 128, 128, 128, 8):
        o, (a, b_, i_) = eng.forward(x[at:at + c], decode_mode=1)
        assert torch.equal(got[at:at + c], o) and torch.equal(xy[at:at + c], a) and torch.equal(idx[at:at + c], i_)
        at += c
    assert torch.isfinite(got).all()


def test_lifter_with_apply_dropout_takes_the_torch_graph_not_the_hip_program():
    """[round 6] trainer.py:424-428: eval mode with the dropout layers switched back to train mode.  The HIP program has
    no dropout; the module must not run it silently (launch counter unchanged, masks drawn: two calls differ); back in
    plain eval mode the HIP program runs again."""
    from egonet_amd import _lib
    cfg = configs.tiny_config()
    cfg['FCModel']['dropout'] = 0.5
    net = hip_fc.get_fc_model(1, cfg, 66, 96)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=3))
    net = net.eval().cuda()
    x = torch.randn(8, 66, generator=torch.Generator().manual_seed(1)).cuda()
    L = _lib.lib()
    n0 = L.egn_launch_count()
    y0 = net(x)
    assert L.egn_launch_count() > n0                        # the HIP program
    for m in net.modules():
        if type(m) == torch.nn.Dropout:
            m.train()
    n1 = L.egn_launch_count()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert L.egn_launch_count() == n1 and not torch.equal(a, b)
    net.eval()
    assert torch.equal(net(x), y0) and L.egn_launch_count() > n1


def test_layer1_on_the_pw_pair_kernel_equals_the_layerwise_program(monkeypatch):
    """[round 5] engine._layer1: conv3 + residual + ReLU + the next block's conv1 + ReLU as ONE launch of csrc/conv_pw.hip
    (the downsample conv and the last conv3 its one-product form) against the same engine with EGONET_AMD_PW_FUSE=0
    (one general conv launch per layer): 3 launches fewer, 5 pw launches, outputs equal to fp32 rounding (the BatchNorm
    scale is folded into the 1x1 filters instead of applied in the epilogue)."""
    cfg = configs.w48_config('heatmap')
    net, sd = _model(cfg, 9)
    x = synth.synth_crops(2, 3, 256, 256, seed=31).cuda()
    outs, kinds = [], []
    for fuse in ('0', '1'):
        monkeypatch.setenv('EGONET_AMD_PW_FUSE', fuse)
        net._engine = None
        eng = net._hip_engine()
        outs.append(eng.forward(x).cpu())
        prog = eng.program(x)
        kinds.append([m['kind'] for m in prog.meta if m['kind'] not in ('fork', 'join')])
    assert kinds[0].count('pwpair') == 0 and kinds[1].count('pwpair') == 5
    assert len(kinds[0]) - len(kinds[1]) == 3        # 4 x conv3 + 3 x conv1 -> 3 pairs + the last conv3
    scale = float(outs[0].abs().max())
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5 * scale

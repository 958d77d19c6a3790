"""Host logic of egonet_amd.trainer that needs no GPU: how the reference's loss object /
config is read into the native step's loss weights (train_IGRs.py:29-46, function.py:49-93)."""
import pytest
import torch.nn as nn

from egonet_amd import trainer


class _Loss(object):                      # the attributes JointsCompositeLoss carries
    def __init__(self, comp, **kw):
        self.comp_dict = comp
        self.__dict__.update(kw)


def test_shipped_config_cross_ratio_off():
    cfgs = {'heatmapModel': {'loss_spec_list': ['mse', 'l1', 'sl1'], 'loss_weight_list': [1.0, 0.1, 'None'],
                             'cr_loss_threshold': 0.15}}
    assert trainer._loss_weights(None, cfgs) == (1.0, 0.1, {})
    comp = {'hm': (nn.MSELoss(), 1.0), 'coor': (nn.L1Loss(), 0.1), 'cr': (nn.SmoothL1Loss(), 'None')}
    assert trainer._loss_weights(_Loss(comp), cfgs) == (1.0, 0.1, {})


def test_cross_ratio_term_is_read_from_the_loss_object():
    comp = {'hm': (nn.MSELoss(), 1.0), 'coor': (nn.L1Loss(), 0.1), 'cr': (nn.SmoothL1Loss(), 0.01)}
    lf = _Loss(comp, cr_indices=[[1, 9, 21, 2]], target_cr=4 / 3, cr_loss_thres=0.1)
    w_hm, w_coor, cr = trainer._loss_weights(lf, {'heatmapModel': {}})
    assert (w_hm, w_coor) == (1.0, 0.1)
    assert cr == dict(w_cr=0.01, cr_type='sl1', cr_indices=[[1, 9, 21, 2]], target_cr=4 / 3, cr_loss_thres=0.1)


def test_cross_ratio_term_from_the_config_alone():
    cfgs = {'heatmapModel': {'loss_spec_list': ['mse', 'None', 'mse'], 'loss_weight_list': [2.0, 0.1, 0.5],
                             'cr_loss_threshold': 0.2}}
    w_hm, w_coor, cr = trainer._loss_weights(None, cfgs)
    assert (w_hm, w_coor) == (2.0, 0.0)
    assert cr['w_cr'] == 0.5 and cr['cr_type'] == 'mse' and cr['cr_loss_thres'] == 0.2 and cr['cr_indices'] is None
    assert abs(cr['target_cr'] - 4 / 3) < 1e-15


def test_every_criterion_of_loss_dict_is_read():
    """function.py:17-20: mse / l1 / sl1 for each of the three terms; anything else is refused."""
    comp = {'hm': (nn.SmoothL1Loss(), 1.0), 'coor': (nn.MSELoss(), 0.3)}
    assert trainer._loss_weights(_Loss(comp), {}) == (1.0, 0.3, dict(hm_type='sl1', coor_type='mse'))
    w_hm, w_coor, cr = trainer._loss_weights(None, {'heatmapModel': {'loss_spec_list': ['l1', 'sl1', 'None'],
                                                                    'loss_weight_list': [1.0, 0.1, 'None']}})
    assert (w_hm, w_coor, cr) == (1.0, 0.1, dict(hm_type='l1', coor_type='sl1'))
    with pytest.raises(NotImplementedError):
        trainer._loss_weights(_Loss({'hm': (nn.HuberLoss(), 1.0)}), {})
    with pytest.raises(NotImplementedError):
        trainer._loss_weights(None, {'heatmapModel': {'loss_spec_list': ['mse', 'huber', 'None'],
                                                      'loss_weight_list': [1.0, 0.1, 'None']}})


def test_optimizer_families_of_the_reference_config():
    """optimizer.py:8-40: adam / sgd with weight_decay, momentum; the native update takes its hyper-parameters
    from the torch optimizer object the caller keeps for the schedule."""
    import torch
    net = nn.Linear(3, 2)
    o = dict(lr=0.01, weight_decay=1e-4, momentum=0.9, milestones=[2], gamma=0.5)
    optim, sche = trainer.prepare_optim(net, {'optimizer': dict(o, optim_type='sgd')})
    assert isinstance(optim, torch.optim.SGD) and isinstance(sche, torch.optim.lr_scheduler.MultiStepLR)
    assert trainer._optim_kwargs(optim, {}) == dict(optim_type='sgd', momentum=0.9, weight_decay=1e-4)
    optim, _ = trainer.prepare_optim(net, {'optimizer': dict(o, optim_type='adam')})
    kw = trainer._optim_kwargs(optim, {})
    assert kw['optim_type'] == 'adam' and kw['weight_decay'] == 1e-4 and kw['betas'] == (0.9, 0.999) and kw['eps'] == 1e-8
    assert trainer._optim_kwargs(None, {'optimizer': dict(o, optim_type='sgd')}) == \
        dict(optim_type='sgd', momentum=0.9, weight_decay=1e-4)
    with pytest.raises(NotImplementedError):
        trainer.prepare_optim(net, {'optimizer': dict(o, optim_type='rmsprop')})
    with pytest.raises(NotImplementedError):
        trainer._optim_kwargs(torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True), {})

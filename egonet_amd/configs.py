"""Config dictionaries in the reference's YAML schema.

The hot path reads only ``cfgs['heatmapModel']`` and ``cfgs['FCModel']``
(reference: libs/model/heatmapModel/hrnet.py:311-469,675-690 and
libs/model/FCmodel.py:107-121).  These builders produce dictionaries that are
key-for-key what ``yaml.safe_load`` gives for the shipped files
(configs/KITTI_inference:demo.yml:60-151, configs/KITTI_train_IGRs_Ped.yml:72-156)
so the same dict drives the reference, the oracle and this build.
"""
import copy


def _stage(num_modules, channels, num_blocks=4, block='basic'):
    nb = len(channels)
    return {'num_modules': num_modules, 'num_branches': nb, 'block': block,
            'num_blocks': [num_blocks] * nb, 'num_channels': list(channels),
            'fuse_method': 'sum'}


def hrnet_config(width=48, input_size=(256, 256), num_joints=33,
                 head_type='coordinates', modules=(1, 4, 3), num_blocks=4,
                 lifter_neurons=1024):
    """input_size is [width, height] like the YAML."""
    w = width
    iw, ih = input_size
    return {
        'FCModel': {'name': 'lifter', 'refine_3d': False, 'norm_twoD': False,
                    'num_blocks': 2, 'input_size': num_joints * 2,
                    'output_size': (num_joints - 1) * 3,
                    'num_neurons': lifter_neurons, 'dropout': 0.5, 'leaky': False},
        'heatmapModel': {
            'name': 'hrnet', 'add_xy': False, 'input_size': [iw, ih],
            'head_type': head_type, 'pixel_shuffle': False,
            'heatmap_size': [iw // 4, ih // 4], 'init_weights': True,
            'pretrained': '', 'num_joints': num_joints,
            'extra': {
                'pretrained_layers': ['conv1', 'bn1', 'conv2', 'bn2', 'layer1',
                                      'transition1', 'stage2', 'transition2',
                                      'stage3', 'transition3', 'stage4'],
                'final_conv_kernel': 1,
                'stage2': _stage(modules[0], [w, 2 * w], num_blocks),
                'stage3': _stage(modules[1], [w, 2 * w, 4 * w], num_blocks),
                'stage4': _stage(modules[2], [w, 2 * w, 4 * w, 8 * w], num_blocks),
            },
        },
        'testing_settings': {'alpha_mode': 'proj'},
    }


def w48_config(head_type='coordinates'):
    """HRNet-W48, 256x256, 33 joints: the demo.yml model."""
    return hrnet_config(48, (256, 256), 33, head_type)


def ped_config(head_type='coordinates'):
    """W32, 192x256 (W x H) input: the Pedestrian training config."""
    return hrnet_config(32, (192, 256), 33, head_type)


def tiny_config(head_type='coordinates', input_size=(64, 64), width=8,
                num_joints=5):
    """Small net with the full topology (fixtures / fast tests)."""
    return hrnet_config(width, input_size, num_joints, head_type,
                        modules=(1, 1, 1), num_blocks=1, lifter_neurons=128)


def clone(cfg):
    return copy.deepcopy(cfg)

"""Validation metrics of the key-point model with the reference's names and return
values (``libs/metric/criterions.py``): ``get_distance`` :17-37, ``get_PCK`` :57-66,
``get_distance_src`` :68-143, ``get_angle_error`` :39-55, ``JointDistance2DSIP``
:173-224, ``AngleError`` :145-171.

The heat-map decode inside ``get_distance_src`` runs on the GPU (csrc/decode.hip, one
wavefront per map: hard arg-max, soft-arg-max, or the numpy-style soft-arg-max); the
rest -- 33 points per instance through a 2x3 inverse crop affine, distances, PCK
counts -- is per-instance host arithmetic in float64 like the reference's.
"""
import numpy as np
import torch

from ..common import img_proc as lip

PCK_THRES = np.array([0.1, 0.2, 0.3])


def get_distance(gt, pred):
    """Per-joint Euclidean distances as a list; a third ground-truth column is a
    visibility flag (zero = joint skipped)."""
    gt = np.asarray(gt)
    if gt.shape[1] not in (2, 3):
        raise ValueError('Array shape not supported.')
    dist = np.sqrt(((gt[:, :2] - pred) ** 2).sum(axis=1))
    if gt.shape[1] == 3:
        dist = dist[np.nonzero(gt[:, 2])[0]]
    return list(dist)


def get_PCK(pred, gt):
    """Counts of key-points closer than PCK_THRES x (a third of the instance's vertical
    extent).  Argument order as in the reference: (prediction, ground truth)."""
    distance = np.array(get_distance(gt, pred))
    denominator = (gt[:, 1].max() - gt[:, 1].min()) / 3
    return np.array([float((distance < t * denominator).sum()) for t in PCK_THRES])


def get_angle_error(pred, meta_data, cfgs=None):
    if not isinstance(pred, np.ndarray):
        pred = pred.data.cpu().numpy()
    dif = np.abs(meta_data['angles_gt'] - np.arctan2(pred[:, 1], pred[:, 0])) * 180 / np.pi
    dif = np.where(dif > 180, 360 - dif, dif)
    return dif.sum() / len(pred), len(pred), None


def _decode(output, arg_max):
    """(local coordinates [N,K,2] numpy, maxvals or None, heat-map width or None)."""
    if type(output) is tuple:                         # (maps, coords in [0,1]): the coordinate head
        return output[1].data.cpu().numpy().astype(np.float32), None, None
    if isinstance(output, torch.Tensor) and not output.is_cuda:
        output = output.cuda()                        # the decode kernels run on the device
    if isinstance(output, np.ndarray) and arg_max == 'soft':
        pred, mv = lip.soft_arg_max_np(output)
        return pred, mv, output.shape[3]
    if isinstance(output, torch.Tensor) and arg_max == 'soft':
        pred, mv = lip.soft_arg_max(output)
        return pred.cpu().numpy(), mv.cpu().numpy(), output.shape[3]
    if isinstance(output, np.ndarray) or (isinstance(output, torch.Tensor) and arg_max == 'hard'):
        pred, mv = lip.get_max_preds(output)          # numpy in -> numpy out; CUDA tensor -> CUDA tensors
        if torch.is_tensor(pred):
            pred, mv = pred.cpu().numpy(), mv.cpu().numpy()
        return pred, mv, output.shape[3]
    raise NotImplementedError


def get_distance_src(output, meta_data, cfgs=None, image_size=(256.0, 256.0), arg_max='hard'):
    """Mean pixel distance, in the SOURCE image, between predicted and annotated
    key-points: decode -> rescale to the crop resolution -> inverse crop affine
    (centre / scale / rotation from meta_data) -> distances + PCK counts.
    Returns (avg distance, number of joints counted, dict of by-products)."""
    pred, max_vals, map_w = _decode(output, arg_max)
    image_size = image_size if cfgs is None else cfgs['heatmapModel']['input_size']
    width, height = image_size
    if map_w is None:
        pred = pred * np.array(image_size).reshape(1, 1, 2)
    else:
        pred = pred * (image_size[0] / map_w)
    centers, scales = meta_data['center'], meta_data['scale']
    used = pred[:len(centers)]                        # extra predictions belong to unlabeled data
    rots = meta_data['rotation'] if 'rotation' in meta_data else [0.] * len(centers)
    originals = meta_data['original_joints']
    distances, correct, src_all = [], np.zeros(len(PCK_THRES)), []
    for i in range(len(used)):
        t_inv = lip.get_affine_transform(centers[i], scales[i], rots[i], (height, width), inv=1)
        src = lip.affine_transform_modified(used[i], t_inv)
        src_all.append(src[None])
        gt = np.asarray(originals[i])
        distances += get_distance(gt, src)
        correct += get_PCK(src, gt)
    cnt = len(distances)
    others = {'src_coord': np.concatenate(src_all, axis=0), 'joints_pred': pred, 'max_vals': max_vals,
              'correct_cnt': correct, 'PCK_batch': correct / cnt}
    return sum(distances) / cnt, cnt, others


class AngleError(object):
    def __init__(self, cfgs, num_joints=None):
        self.name = 'Angle error in degrees'
        self.num_joints, self.count, self.mean = num_joints, 0, 0.

    def update(self, prediction, meta_data, ground_truth=None, logger=None):
        avg, cnt, _ = get_angle_error(prediction, meta_data)
        self.mean = (self.mean * self.count + cnt * avg) / (self.count + cnt)
        self.count += cnt

    def report(self, logger):
        logger.info('Error type: {:s}\tError: {}\t'.format(self.name, self.mean))


class JointDistance2DSIP(object):
    """Running mean of get_distance_src + PCK over an evaluation pass."""

    def __init__(self, cfgs, num_joints=None):
        self.name = 'Joint distance in the source image plane'
        self.num_joints = num_joints if num_joints is not None else cfgs['heatmapModel']['num_joints']
        self.image_size = cfgs['heatmapModel']['input_size']
        self.arg_max = cfgs['testing_settings'].get('arg_max')
        self.count, self.mean, self.PCK_counts = 0, 0., np.zeros(len(PCK_THRES))

    def update(self, prediction, meta_data, ground_truth=None, logger=None):
        avg, cnt, others = get_distance_src(prediction, meta_data, arg_max=self.arg_max, image_size=self.image_size)
        self.mean = (self.mean * self.count + cnt * avg) / (self.count + cnt)
        self.count += cnt
        self.PCK_counts += others['correct_cnt']

    def report(self, logger):
        logger.info('Ealuaton Results:')
        logger.info('Error type: {:s}\tMPJPE: {}\t'.format(self.name, self.mean))
        for thres, value in zip(PCK_THRES, self.PCK_counts):
            logger.info('PCK at threshold {:.2f}: {:.3f}'.format(thres, value / self.count))

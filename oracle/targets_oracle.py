"""CPU oracle: heat-map training targets (TEST INFRASTRUCTURE ONLY).

What libs/common/img_proc.py:347-409 ``generate_target`` computes, stated per
heat-map pixel instead of per pasted patch: joint j puts
    exp(-((px - ulx - c)^2 + (py - uly - c)^2) / (2 sigma^2))        (float32)
on the pixels of the window [ul, br) clipped to the map, where
    mu = int(joint / stride + 0.5),  ul = int(mu - 3 sigma),  br = int(mu + 3 sigma + 1),
    c = (6 sigma + 1) // 2,  stride = input_size / heatmap_size (element-wise),
and a joint whose window misses the map entirely gets weight 0 (:382-386).
Pinned by tests/test_oracle_golden.py against the reference's own function
(tests/golden/targets.npz).
"""
import numpy as np


def generate_target(joints, joints_vis, input_size, heatmap_size, sigma):
    """joints [K,>=2], joints_vis [K] -> (target [K,hs[0],hs[1]] f32, target_weight [K,1] f32)."""
    k = len(joints)
    rows, cols = int(heatmap_size[0]), int(heatmap_size[1])
    sx = float(input_size[0]) / float(heatmap_size[0])
    sy = float(input_size[1]) / float(heatmap_size[1])
    half = sigma * 3
    side = 2 * half + 1
    centre = np.float32(side // 2)
    glen = len(np.arange(0, side, 1))
    weight = np.asarray(joints_vis, dtype=np.float32).reshape(k, 1).copy()
    target = np.zeros((k, rows, cols), dtype=np.float32)
    py, px = np.mgrid[0:rows, 0:cols]
    for j in range(k):
        if not weight[j, 0] > 0.5:
            continue
        mu = int(joints[j][0] / sx + 0.5), int(joints[j][1] / sy + 0.5)
        ul = int(mu[0] - half), int(mu[1] - half)
        br = int(mu[0] + half + 1), int(mu[1] + half + 1)
        if ul[0] >= cols or ul[1] >= rows or br[0] < 0 or br[1] < 0:
            weight[j, 0] = 0
            continue
        gx, gy = px - ul[0], py - ul[1]
        window = (gx >= 0) & (gy >= 0) & (gx < glen) & (gy < glen) & (px < min(br[0], cols)) & (py < min(br[1], rows))
        d2 = (gx.astype(np.float32) - centre) ** 2 + (gy.astype(np.float32) - centre) ** 2
        dot = np.exp(-(d2 / np.float32(2 * sigma ** 2))).astype(np.float32)
        target[j] = np.where(window, dot, np.float32(0))
    return target, weight


def generate_target_batch(joints, joints_vis, input_size, heatmap_size, sigma):
    outs = [generate_target(joints[i], joints_vis[i], input_size, heatmap_size, sigma) for i in range(len(joints))]
    return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])

"""GPU: rdfc_nyu_prepare_batch / rdfc_gan_amd.batch_prep.NYUv2BatchPrep against the oracle (oracle/batch_prep.py, pinned against
PIL in test_batch_prep_cpu.py): every entry of the sample dict of nyuv2_dataset_training.py:75-155, bit for bit."""
import numpy as np
import pytest
import torch

from test_batch_prep_cpu import _sample

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_batch_prep_is_bit_identical_to_the_host_chain():
    from oracle import batch_prep as bp
    from rdfc_gan_amd.batch_prep import NYUv2BatchPrep
    angles = [3.7, -4.99, 0.31, -2.2]
    samples = [_sample(10 + i) for i in range(len(angles))]
    prep = NYUv2BatchPrep(DEV)
    stack = lambda k: np.stack([s[k] for s in samples])
    got = prep(stack(0), stack(1), stack(2), stack(3), stack(4), angles=angles)
    assert got['rgb'].shape == (4, 3, 224, 304) and got['labels'].dtype == torch.int64 and got['depth_masks'].dtype == torch.bool
    assert got['normal_masks'].shape == (4, 224, 304) and got['raw_depth'].shape == (4, 1, 224, 304)
    for i, (s, a) in enumerate(zip(samples, angles)):
        ref = bp.prepare_sample(*s, a)
        for k, v in ref.items():
            g = got[k][i].cpu().numpy()
            assert g.shape == v.shape, (k, g.shape, v.shape)
            assert np.array_equal(g, v), f'sample {i} {k}: {int((g != v).sum())} of {v.size} elements differ, max {np.abs(g.astype(np.float64) - v.astype(np.float64)).max()}'


def test_angles_come_from_numpys_stream_like_upstream_and_feed_the_model():
    from rdfc_gan_amd.batch_prep import NYUv2BatchPrep
    prep = NYUv2BatchPrep(DEV)
    np.random.seed(7)
    mine = prep.draw_angles(3)
    np.random.seed(7)
    assert mine == [float(np.random.uniform(-5.0, 5.0)) for _ in range(3)]
    s = [_sample(20 + i) for i in range(2)]
    stack = lambda k: torch.from_numpy(np.stack([x[k] for x in s]))
    np.random.seed(8)
    batch = prep(stack(0), stack(1), stack(2), stack(3), stack(4))
    assert set(batch) == {'rgb', 'gt_normal', 'raw_depth', 'gt_depth', 'depth_masks', 'normal_masks', 'labels'}
    assert all(v.is_cuda for v in batch.values()) and float(batch['gt_depth'].min()) >= -1.0 - 1e-6 and float(batch['gt_depth'].max()) <= 1.0

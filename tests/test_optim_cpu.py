"""FlatAdam <-> torch.optim.Adam / AdamW checkpoint exchange (lib/utils/checkpoint.py:21-58 stores optimizer.state_dict()).
Host logic only: the update itself is rdfc_adam_step (HIP) and is covered by the GPU tests."""
import copy

import pytest
import torch
import torch.nn as nn

from rdfc_gan_amd.optim import FlatAdam


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Conv2d(4, 2, 1))


@pytest.mark.parametrize('wd', [None, 0.01])
def test_state_dict_loads_into_torch_optimizer_and_steps(wd):
    net = _net()
    opt = FlatAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.999), weight_decay=wd)
    opt.steps = 3
    opt.exp_avg.normal_()
    opt.exp_avg_sq.uniform_(0.1, 1.0)
    sd = opt.state_dict()
    assert isinstance(sd['param_groups'][0]['weight_decay'], float)          # never None: torch's step() does arithmetic on it
    ref_net = copy.deepcopy(net)
    ref = (torch.optim.AdamW if wd is not None else torch.optim.Adam)(ref_net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    ref.load_state_dict(sd)
    for p in ref_net.parameters():
        p.grad = torch.ones_like(p)
    ref.step()                                                               # raised TypeError on alpha=None in round 1
    st = ref.state_dict()['state']
    assert float(st[0]['step']) == 4.0
    # the moments land on the right parameters, in logical (OIHW) order although FlatAdam stores conv weights channels-last
    p0 = list(net.parameters())[0]
    assert st[0]['exp_avg'].shape == p0.shape


def test_loading_a_torch_adam_checkpoint_keeps_the_update_rule():
    net = _net()
    ref = torch.optim.Adam(net.parameters(), lr=2e-3, betas=(0.5, 0.999))
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    ref.step()
    sd = ref.state_dict()
    assert sd['param_groups'][0]['weight_decay'] == 0
    mine = FlatAdam(_net().parameters(), lr=1e-3)
    mine.load_state_dict(sd)
    assert mine._adamw is False and mine.steps == 1                           # upstream's `weight_decay: 0` does not select AdamW
    assert mine.param_groups[0]['lr'] == 2e-3
    got = mine.state_dict()['state'][0]['exp_avg']
    assert torch.equal(got, sd['state'][0]['exp_avg'])
    sd2 = copy.deepcopy(sd)
    sd2['param_groups'][0]['weight_decay'] = 0.1
    with pytest.raises(ValueError):
        mine.load_state_dict(sd2)


def test_round1_checkpoint_with_none_weight_decay_still_loads():
    mine = FlatAdam(_net().parameters())
    sd = mine.state_dict()
    sd['param_groups'][0]['weight_decay'] = None
    mine.load_state_dict(sd)
    assert mine.param_groups[0]['weight_decay'] == 0.0

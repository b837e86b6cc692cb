"""Oracle train step: RDFCGAN.optimize_parameters restricted to the hot-path nets.

TEST INFRASTRUCTURE ONLY.  Restates the call order of
RDFC-GAN/lib/models/rdfc_gan.py: set_input :66-77, forward :184-207,
backward_G :209-289, backward_D_basic :292-300, optimize_parameters :318-340.
G_normal / G_label are outside the hot path: the surface normals they would
produce are inputs here (`normal_fn`), and the label / normal / manhattan loss
terms are omitted (they do not depend on the hot-path nets when G_normal and
G_label outputs are held fixed).
"""
import itertools

import torch

from .losses import GANLoss, l1_loss


class HotPathStep:
    def __init__(self, G_A2B, G_B2A, D_A, D_B, lr=2e-3, betas=(0.5, 0.999),
                 lambda_A=100.0, lambda_B=100.0, lambda_L1=100.0, gan_mode='lsgan'):
        self.G_A2B, self.G_B2A, self.D_A, self.D_B = G_A2B, G_B2A, D_A, D_B
        self.lam = (lambda_A, lambda_B, lambda_L1)
        self.gan = GANLoss(gan_mode)
        self.opt_G = torch.optim.Adam(itertools.chain(G_A2B.parameters(), G_B2A.parameters()), lr=lr, betas=betas)
        self.opt_D = torch.optim.Adam(itertools.chain(D_A.parameters(), D_B.parameters()), lr=lr, betas=betas)

    def set_input(self, data):
        self.real_A, self.real_B, self.aux_A = data['rgb'], data['gt_depth'], data['raw_depth']
        mask = data['depth_masks'] if 'depth_masks' in data else torch.ones_like(self.real_B)
        self.w = mask / (mask.sum() + 1e-6)                                   # rdfc_gan.py:77

    @staticmethod
    def _req(nets, flag):
        for n in nets:
            for p in n.parameters():
                p.requires_grad = flag

    def step(self, normal_fn, pepper_fn=None):
        """normal_fn(tag) -> unit normals (B,3,H,W) for tag in
        ('real_A', 'fake_A', 'fake_A_again'); pepper_fn(fake_B) mutates in place
        (helper.py:384-402) or is None."""
        lamA, lamB, lam1 = self.lam
        # ---- forward (rdfc_gan.py:184-207)
        fake_B = self.G_A2B(self.real_A, self.aux_A, normal_fn('real_A'))['pred_depth']
        if pepper_fn is not None:
            fake_B = pepper_fn(fake_B)
        self.G_B2A(fake_B)                                                     # dead forward :197
        fake_A = self.G_B2A(self.real_B)
        self.G_A2B(fake_A, self.aux_A, normal_fn('fake_A'))                    # dead forward :207
        # ---- G step (:321-329)
        self._req([self.D_A, self.D_B], False)
        self.opt_G.zero_grad()
        L = {}
        L['loss_G_A2B'] = self.gan(self.D_A(fake_B), True)
        L['loss_G_B2A'] = self.gan(self.D_B(fake_A), True)
        a2b_l1 = l1_loss(fake_B, self.real_B, self.w) * lam1 * 5
        b2a_l1 = l1_loss(fake_A, self.real_A, self.w) * lam1 * 3
        rec_A = self.G_B2A(fake_B)
        L['loss_cycle_A'] = l1_loss(rec_A, self.real_A) * lamA
        rec_B = self.G_A2B(fake_A, self.aux_A, normal_fn('fake_A_again'))['pred_depth']
        L['loss_cycle_B'] = l1_loss(rec_B, self.real_B, self.w) * lamB
        loss_G = L['loss_G_A2B'] + L['loss_cycle_A'] + L['loss_cycle_B'] + L['loss_G_B2A'] + a2b_l1 + b2a_l1
        loss_G.backward()
        L['loss_G'] = loss_G
        L['loss_B2A_L1'] = b2a_l1
        L['loss_A2B_L1'] = b2a_l1           # reference logs B2A under this key (rdfc_gan.py:276)
        L['loss_A2B_L1_true'] = a2b_l1
        self.opt_G.step()
        # ---- D step (:331-335)
        self._req([self.D_A, self.D_B], True)
        self.opt_D.zero_grad()
        for tag, D, real, fake in (('A', self.D_A, self.real_B, fake_B), ('B', self.D_B, self.real_A, fake_A)):
            lr_ = self.gan(D(real), True)
            lf_ = self.gan(D(fake.detach()), False)
            ld_ = (lr_ + lf_) * 0.5
            ld_.backward()
            L[f'loss_D_{tag}'], L[f'loss_D_{tag}_real'], L[f'loss_D_{tag}_fake'] = ld_, lr_, lf_
        self.opt_D.step()
        self.fake_B, self.fake_A = fake_B.detach(), fake_A.detach()
        return {k: float(v.detach()) for k, v in L.items()}

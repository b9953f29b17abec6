"""Hierarchical VAE-GAN on the MI355X HIP kernels — drop-in for hierarchical_vae_gan/model.py:1-436.

Same constructors, forward signatures and state_dict keys (267 tensors) as the reference classes.
The reference's composed ``HierarchicalDecoder.forward`` raises (``upsample[i-1]`` is indexed
shallow->deep, model.py:234-238,261); the default here raises the same way.  The NON-REFERENCE
composition D2 of SURVEY.md §8(c) (``hier_decoder_policy='nearest'``: sub-blocks chained with
nearest x2 upsampling, ``decoder.upsample.*`` unused) exists only so that a whole step can run.
"""
import torch
import torch.nn as nn

from .. import lib as L
from .. import ops
from ..nn import ResidualBlock, FusedSequential, Conv2d, ConvTranspose2d, BatchNorm2d, LeakyReLU, Sigmoid


class SelfAttention(nn.Module):
    """hierarchical_vae_gan/model.py:12-47 on NHWC activations: 1x1 q/k/v convs, fused flash-style
    softmax(q k^T) v without 1/sqrt(d), out = gamma*att + x."""

    def __init__(self, in_channels):
        super().__init__()
        self.query = Conv2d(in_channels, in_channels // 8, 1)
        self.key = Conv2d(in_channels, in_channels // 8, 1)
        self.value = Conv2d(in_channels, in_channels, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return ops.attention(self.query(x), self.key(x), self.value(x), x, self.gamma)


def _level_block(cin, cout, n_res, attn):
    blk = FusedSequential(Conv2d(cin, cout, 4, stride=2, padding=1) if n_res == 2 else Conv2d(cin, cout, 3, padding=1),
                          BatchNorm2d(cout), LeakyReLU(0.2), *[ResidualBlock(cout, cout) for _ in range(n_res)])
    return FusedSequential(blk, SelfAttention(cout)) if attn else blk


class HierarchicalEncoder(nn.Module):
    """hierarchical_vae_gan/model.py:83-173."""

    def __init__(self, input_channels=3, base_channels=64, latent_dims=[32, 64, 128], num_hierarchies=3):
        super().__init__()
        assert len(latent_dims) == num_hierarchies, "Number of latent dimensions must match number of hierarchies"
        self.num_hierarchies, self.latent_dims = num_hierarchies, list(latent_dims)
        self.init_conv = FusedSequential(Conv2d(input_channels, base_channels, 3, padding=1), BatchNorm2d(base_channels),
                                         LeakyReLU(0.2))
        self.encoders, self.mu_layers, self.logvar_layers = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        cur = base_channels
        for i in range(num_hierarchies):
            lc = base_channels * 2 ** i
            self.encoders.append(_level_block(cur, lc, 2, i >= 1))
            self.mu_layers.append(Conv2d(lc, latent_dims[i], 1))
            self.logvar_layers.append(Conv2d(lc, latent_dims[i], 1))
            cur = lc

    def forward(self, x):
        mu_list, logvar_list = [], []
        x = self.init_conv(ops.to_nhwc(x))
        for i in range(self.num_hierarchies):
            x = self.encoders[i](x)
            mu_list.append(ops.to_nchw(self.mu_layers[i](x), self.latent_dims[i]))
            logvar_list.append(ops.clamp(ops.to_nchw(self.logvar_layers[i](x), self.latent_dims[i]), -20, 20))
        return mu_list, logvar_list


class HierarchicalDecoder(nn.Module):
    """hierarchical_vae_gan/model.py:176-272."""

    def __init__(self, output_channels=3, base_channels=64, latent_dims=[32, 64, 128], num_hierarchies=3):
        super().__init__()
        assert len(latent_dims) == num_hierarchies, "Number of latent dimensions must match number of hierarchies"
        self.num_hierarchies, self.latent_dims = num_hierarchies, list(latent_dims)
        self.output_channels, self.base_channels = output_channels, base_channels
        self.policy = 'error'
        self.decoders = nn.ModuleList()
        for i in range(num_hierarchies - 1, -1, -1):
            lc = base_channels * 2 ** i
            cin = latent_dims[i] if i == num_hierarchies - 1 else latent_dims[i] + base_channels * 2 ** (i + 1)
            self.decoders.append(_level_block(cin, lc, 1, i >= 1))
        self.output_conv = FusedSequential(Conv2d(base_channels, base_channels, 3, padding=1), BatchNorm2d(base_channels),
                                           LeakyReLU(0.2), Conv2d(base_channels, output_channels, 3, padding=1), Sigmoid())
        self.upsample = nn.ModuleList([ConvTranspose2d(base_channels * 2 ** i, base_channels * 2 ** (i - 1), 4, stride=2, padding=1)
                                       for i in range(1, num_hierarchies)])

    def run_block(self, j, x_nchw):
        """One entry of ``decoders`` on an NCHW tensor (module-level parity target)."""
        lc = self.base_channels * 2 ** (self.num_hierarchies - 1 - j)
        return ops.to_nchw(self.decoders[j](ops.to_nhwc(x_nchw)), lc)

    def run_output_conv(self, x_nchw):
        return ops.to_nchw(self.output_conv(ops.to_nhwc(x_nchw)), self.output_channels)

    def forward(self, z_list):
        x = None
        for j, z in enumerate(reversed(z_list)):
            if j == 0:
                x = self.decoders[0](ops.to_nhwc(z))
                continue
            if self.policy != 'nearest':
                up = self.upsample[j - 1]
                # the reference feeds a (base*2^(nh-j))-channel map into upsample[j-1] built for base*2^j (model.py:261)
                have = self.base_channels * 2 ** (self.num_hierarchies - j)
                raise RuntimeError(f"Given transposed=1, weight of size {list(up.weight.shape)}, expected input"
                                   f"[{x.shape[0]}, {have}, {x.shape[1]}, {x.shape[2]}] to have {up.in_channels} channels, "
                                   f"but got {have} channels instead")
            zc = ops.to_nhwc(z)
            if zc.shape[-1] != z.shape[1]:
                raise RuntimeError("nearest composition needs latent widths that are multiples of the channel granule")
            xu = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)     # nearest x2 (layout plumbing, NHWC)
            x = self.decoders[j](torch.cat([zc, xu], dim=-1))
        return ops.to_nchw(self.output_conv(x), self.output_channels)


class Discriminator(nn.Module):
    """hierarchical_vae_gan/model.py:275-313."""

    def __init__(self, input_channels=3, base_channels=64, num_hierarchies=3):
        super().__init__()
        layers = [Conv2d(input_channels, base_channels, 4, stride=2, padding=1), LeakyReLU(0.2)]
        cur = base_channels
        for i in range(1, num_hierarchies):
            nxt = base_channels * 2 ** i
            layers += [Conv2d(cur, nxt, 4, stride=2, padding=1), BatchNorm2d(nxt), LeakyReLU(0.2), SelfAttention(nxt)]
            cur = nxt
        layers += [Conv2d(cur, cur * 2, 4, stride=1, padding=1), BatchNorm2d(cur * 2), LeakyReLU(0.2),
                   Conv2d(cur * 2, 1, 4, stride=1, padding=1)]
        self.model = FusedSequential(*layers)

    def forward(self, x):
        return ops.to_nchw(self.model(ops.to_nhwc(x)), 1)


class HierarchicalVAEGAN(nn.Module):
    """hierarchical_vae_gan/model.py:316-436."""

    def __init__(self, config):
        super().__init__()
        self.latent_dims = config.latent_dims
        self.num_hierarchies = config.num_hierarchies
        self.beta = config.beta
        self.level_weights = config.level_weights
        self.input_channels = config.channels
        self.base_channels = config.base_channels
        self.lambda_adv = config.lambda_adv
        if len(self.level_weights) != self.num_hierarchies:
            self.level_weights = [1.0] * self.num_hierarchies
        self.recon_target_policy = getattr(config, 'recon_target_policy', 'error')
        self.encoder = HierarchicalEncoder(self.input_channels, self.base_channels, self.latent_dims, self.num_hierarchies)
        self.decoder = HierarchicalDecoder(self.input_channels, self.base_channels, self.latent_dims, self.num_hierarchies)
        self.decoder.policy = getattr(config, 'hier_decoder_policy', 'error')
        self.discriminator = Discriminator(self.input_channels, self.base_channels, self.num_hierarchies)

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)
        return ops.reparameterize(mu, logvar, eps)

    def encode(self, x, eps_list=None):
        mu_list, logvar_list = self.encoder(x)
        eps_list = eps_list or [None] * len(mu_list)
        z_list = [self.reparameterize(m, lv, e) for m, lv, e in zip(mu_list, logvar_list, eps_list)]   # level order 0..n-1
        return z_list, mu_list, logvar_list

    def decode(self, z_list):
        return self.decoder(z_list)

    def forward(self, x, eps_list=None):
        z_list, mu_list, logvar_list = self.encode(x, eps_list)
        return self.decode(z_list), z_list, mu_list, logvar_list

    def loss_function(self, original, recon_x, mu_list, logvar_list, disc_real, disc_fake, train_generator=True):
        target = original
        if original.shape != recon_x.shape and self.recon_target_policy == 'area':
            target = ops.avg_pool2(original)
        recon_loss = ops.mse(recon_x, target)
        kl_loss = 0.0
        for mu, logvar, w in zip(mu_list, logvar_list, self.level_weights):
            kl_loss = kl_loss + w * ops.kl_divergence(mu, logvar)
        if train_generator:
            adv_loss = ops.neg_mean(disc_fake)
            total = recon_loss + self.beta * kl_loss + self.lambda_adv * adv_loss
            return {'total': total, 'recon': recon_loss.detach(), 'kl': kl_loss.detach(), 'adv': adv_loss.detach(), 'disc': 0.0}
        disc_loss = ops.hinge_d(disc_real, disc_fake)
        return {'total': disc_loss, 'recon': recon_loss.detach(), 'kl': kl_loss.detach(), 'adv': 0.0,
                'disc': disc_loss.detach()}

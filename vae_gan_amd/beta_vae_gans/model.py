"""β-VAE-GAN on the MI355X HIP kernels — drop-in for beta_vae_gans/model.py.

Same constructors, forward signatures and state_dict keys as the reference classes.  As in the
reference, the encoder's Linear layers pin the input to 256x256 and the decoder emits 128x128
(model.py:70-75,104,121-126), so ``loss_function`` raises on the MSE shape mismatch exactly like
the reference unless ``recon_target_policy='area'`` (NON-REFERENCE shim D1 of SURVEY.md §8(c)).
"""
import torch
import torch.nn as nn

from .. import ops
from .. import lib as L
from ..nn import ResidualBlock, FusedSequential, Conv2d, BatchNorm2d, LeakyReLU  # noqa: F401
from ..stacks import down_stack, up_stack, patch_disc_stack


class Linear(nn.Module):
    """nn.Linear parameter holder ([out, in] weight + bias), run by the f32 MFMA path."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        bound = 1.0 / in_features ** 0.5
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class Encoder(nn.Module):
    """beta_vae_gans/model.py:43-93."""

    def __init__(self, input_channels=3, hidden_channels=[64, 128, 256, 512], latent_dim=128):
        super().__init__()
        self.features = FusedSequential(*down_stack(input_channels, list(hidden_channels)))
        self.feature_map_size = 256 // (2 ** (len(hidden_channels) - 1))
        self.last_channels = hidden_channels[-1]
        fc_in = hidden_channels[-1] * self.feature_map_size ** 2
        self.fc_mu = Linear(fc_in, latent_dim)
        self.fc_logvar = Linear(fc_in, latent_dim)
        self.grad_boundary = None     # data parallel: callable put on the flattened features (bucket boundary)

    def forward(self, x):
        f = ops.to_nchw(self.features(ops.to_nhwc(x)), self.last_channels)     # NCHW flatten order of the FC weights
        f = f.reshape(x.size(0), -1)
        if f.shape[1] != self.fc_mu.in_features:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({f.shape[0]}x{f.shape[1]} and "
                               f"{self.fc_mu.in_features}x{self.fc_mu.out_features})")
        if self.grad_boundary is not None:
            f = self.grad_boundary(f)
        mu = self.fc_mu(f)
        logvar = ops.clamp(self.fc_logvar(f), -20, 20)
        return mu, logvar


class Decoder(nn.Module):
    """beta_vae_gans/model.py:96-145."""

    def __init__(self, latent_dim=128, hidden_channels=[512, 256, 128, 64], output_channels=3):
        super().__init__()
        self.feature_map_size = 256 // (2 ** len(hidden_channels))
        self.init_size = self.feature_map_size
        self.first_channels, self.output_channels = hidden_channels[0], output_channels
        self.fc = Linear(latent_dim, hidden_channels[0] * self.feature_map_size ** 2)
        self.decoder = FusedSequential(BatchNorm2d(hidden_channels[0]), LeakyReLU(0.2),
                                       *up_stack(list(hidden_channels), output_channels))

    def forward(self, z):
        x = self.fc(z).view(z.size(0), self.first_channels, self.init_size, self.init_size)
        return ops.to_nchw(self.decoder(ops.to_nhwc(x)), self.output_channels)


class Discriminator(nn.Module):
    """beta_vae_gans/model.py:148-176."""

    def __init__(self, input_channels=3, hidden_channels=[64, 128, 256, 512]):
        super().__init__()
        self.model = FusedSequential(*patch_disc_stack(input_channels, list(hidden_channels)))

    def forward(self, x):
        return ops.to_nchw(self.model(ops.to_nhwc(x)), 1)


class BetaVAEGAN(nn.Module):
    """beta_vae_gans/model.py:179-285."""

    def __init__(self, config):
        super().__init__()
        self.latent_dim = config.latent_dim
        self.beta = config.beta
        self.input_channels = config.channels
        self.enc_channels = config.enc_channels
        self.dec_channels = config.dec_channels
        self.lambda_adv = config.lambda_adv
        # 'error' (default) reproduces the reference's failure; 'area' = avg_pool2d(original, 2) target (shim D1)
        self.recon_target_policy = getattr(config, 'recon_target_policy', 'error')
        self.encoder = Encoder(self.input_channels, self.enc_channels, self.latent_dim)
        self.decoder = Decoder(self.latent_dim, self.dec_channels, self.input_channels)
        self.discriminator = Discriminator(self.input_channels, self.enc_channels)

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)          # one draw per call, like model.py:214
        return ops.reparameterize(mu, logvar, eps)

    def encode(self, x, eps=None):
        mu, logvar = self.encoder(x)
        return self.reparameterize(mu, logvar, eps), mu, logvar

    def decode(self, z):
        return self.decoder(z)

    def forward(self, x, eps=None):
        z, mu, logvar = self.encode(x, eps)
        return self.decode(z), z, mu, logvar

    def _target(self, original, recon_x):
        if original.shape != recon_x.shape and self.recon_target_policy == 'area':
            return ops.avg_pool2(original)
        return original

    def loss_function(self, original, recon_x, mu, logvar, disc_real, disc_fake, train_generator=True):
        recon_loss = ops.mse(recon_x, self._target(original, recon_x))      # raises on 128 vs 256 like model.py:255
        kl_loss = ops.kl_divergence(mu, logvar)
        if train_generator:
            adv_loss = ops.neg_mean(disc_fake)
            total = recon_loss + self.beta * kl_loss + self.lambda_adv * adv_loss
            return {'total': total, 'recon': recon_loss.detach(), 'kl': kl_loss.detach(), 'adv': adv_loss.detach(), 'disc': 0.0}
        disc_loss = ops.hinge_d(disc_real, disc_fake)
        return {'total': disc_loss, 'recon': recon_loss.detach(), 'kl': kl_loss.detach(), 'adv': 0.0,
                'disc': disc_loss.detach()}

"""VQ-VAE-GAN on the MI355X HIP kernels — drop-in for vq_vae_gan/model.py:1-333.

Same constructors, forward signatures and state_dict keys as the reference classes
(ResidualBlock, VectorQuantizer, Encoder, Decoder, Discriminator, VQVAEGAN); inputs/outputs are
NCHW float32 tensors on the GPU.  Inside, activations are NHWC bf16 (or f32 in parity mode).
"""
import torch
import torch.nn as nn

from .. import ops
from ..nn import ResidualBlock, FusedSequential, Conv2d  # noqa: F401  (ResidualBlock re-exported)
from ..stacks import down_stack, up_stack, patch_disc_stack


class VectorQuantizer(nn.Module):
    """vq_vae_gan/model.py:43-134: nearest-code search, straight-through output, commitment loss,
    EMA codebook update (driven by the ``training`` argument, not ``module.training``)."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost, decay=0.99):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.commitment_cost, self.decay = commitment_cost, decay
        self.embeddings = nn.Embedding(num_embeddings, embedding_dim)
        self.embeddings.weight.data.uniform_(-1.0 / num_embeddings, 1.0 / num_embeddings)
        self.register_buffer('ema_cluster_size', torch.zeros(num_embeddings))
        self.register_buffer('ema_w', self.embeddings.weight.data.clone())

    def forward(self, inputs, training=True):
        if inputs.shape[1] != self.embedding_dim:
            raise RuntimeError(f"expected {self.embedding_dim} channels, got {inputs.shape[1]}")
        # the codebook is updated in place by the EMA kernel (the reference rebinds .data; the
        # embedding weight is in neither optimizer, vq_vae_gan/main.py:34-44)
        return ops.VectorQuantizeFn.apply(inputs, self.embeddings.weight.data, self.ema_cluster_size, self.ema_w,
                                          float(self.commitment_cost), float(self.decay), bool(training))


class Encoder(nn.Module):
    """vq_vae_gan/model.py:137-166."""

    def __init__(self, input_channels=3, hidden_channels=[64, 128, 256, 512], embedding_dim=64):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.encoder = FusedSequential(*down_stack(input_channels, list(hidden_channels)),
                                       Conv2d(hidden_channels[-1], embedding_dim, 1))

    def forward(self, x):
        return ops.to_nchw(self.encoder(ops.to_nhwc(x)), self.embedding_dim)


class Decoder(nn.Module):
    """vq_vae_gan/model.py:169-199."""

    def __init__(self, embedding_dim=64, hidden_channels=[512, 256, 128, 64], output_channels=3):
        super().__init__()
        from ..nn import BatchNorm2d, LeakyReLU
        self.output_channels = output_channels
        self.decoder = FusedSequential(Conv2d(embedding_dim, hidden_channels[0], 3, padding=1),
                                       BatchNorm2d(hidden_channels[0]), LeakyReLU(0.2),
                                       *up_stack(list(hidden_channels), output_channels))

    def forward(self, x):
        return ops.to_nchw(self.decoder(ops.to_nhwc(x)), self.output_channels)


class Discriminator(nn.Module):
    """vq_vae_gan/model.py:202-226 (PatchGAN logits, no pooling)."""

    def __init__(self, input_channels=3, hidden_channels=[64, 128, 256, 512]):
        super().__init__()
        self.model = FusedSequential(*patch_disc_stack(input_channels, list(hidden_channels)))

    def forward(self, x):
        return ops.to_nchw(self.model(ops.to_nhwc(x)), 1)


class VQVAEGAN(nn.Module):
    """vq_vae_gan/model.py:229-333."""

    def __init__(self, config):
        super().__init__()
        self.embedding_dim = config.embedding_dim
        self.num_embeddings = config.num_embeddings
        self.commitment_cost = config.commitment_cost
        self.decay = config.decay
        self.input_channels = config.channels
        self.enc_channels = config.enc_channels
        self.dec_channels = config.dec_channels
        self.lambda_adv = config.lambda_adv
        self.encoder = Encoder(self.input_channels, self.enc_channels, self.embedding_dim)
        self.vector_quantizer = VectorQuantizer(self.num_embeddings, self.embedding_dim, self.commitment_cost, self.decay)
        self.decoder = Decoder(self.embedding_dim, self.dec_channels, self.input_channels)
        self.discriminator = Discriminator(self.input_channels, self.enc_channels)

    def encode(self, x, training=True):
        z = self.encoder(x)
        z_q, vq_loss, indices = self.vector_quantizer(z, training=training)
        return z_q, z, vq_loss, indices

    def decode(self, z_q):
        return self.decoder(z_q)

    def forward(self, x, training=True):
        z_q, z, vq_loss, indices = self.encode(x, training=training)
        return self.decode(z_q), z, z_q, vq_loss, indices

    def loss_function(self, original, recon_x, vq_loss, disc_real, disc_fake, train_generator=True):
        """Same dict as the reference (model.py:290-333).  'total' carries the graph; the other entries
        are 0-dim device tensors instead of python floats, so no host sync happens inside the step
        (float(x) / x.item() give the reference's values)."""
        recon_loss = ops.mse(recon_x, original)
        zero = 0.0
        if train_generator:
            adv_loss = ops.neg_mean(disc_fake)
            total = recon_loss + vq_loss + self.lambda_adv * adv_loss
            return {'total': total, 'recon': recon_loss.detach(), 'vq': vq_loss.detach(), 'adv': adv_loss.detach(),
                    'disc': zero}
        disc_loss = ops.hinge_d(disc_real, disc_fake)
        return {'total': disc_loss, 'recon': recon_loss.detach(), 'vq': vq_loss.detach(), 'adv': zero,
                'disc': disc_loss.detach()}

"""The distinct fused units of the three architectures, for the teacher-forced per-layer gradient check.

Every unit is built by the SAME expression on both sides — ``build(M)`` with ``M`` the reference's ``model.py``
module (tests/golden/make_golden.py, build container only) or this engine's drop-in module of the same name — which
is the drop-in contract itself: same constructors, same nn.Sequential child order, same state_dict keys.  The unit's
input and its upstream gradient are regenerated from seeds on both sides, its weights from the state_dict key names
(``fill_state_dict(..., salt=unit index)``), so a fixture holds only expected outputs.

A unit has at most two LeakyReLU layers behind each other, so the sign-flip noise that limits the deep-stack goldens
 does not build up: the reference's own fp32 run sits ~1e-6 from its fp64 run on every entry, and
the HIP engine is held to 1e-4 (f32 mode).
"""
from collections import namedtuple

Unit = namedtuple("Unit", "name arch build shape kind")    # kind: 'normal' = N(0,1) input, 'image' = U[0,1)

CH = [64, 128, 256, 512]

UNITS = [
    Unit("conv3_bn_lrelu_3_64", "vq", lambda M: M.Encoder(3, CH, 64).encoder[0:3], (2, 3, 32, 32), "image"),
    Unit("res64", "vq", lambda M: M.ResidualBlock(64, 64), (2, 64, 32, 32), "normal"),
    Unit("down4s2_bn_lrelu_64_128", "vq", lambda M: M.Encoder(3, CH, 64).encoder[4:7], (2, 64, 32, 32), "normal"),
    Unit("res128", "vq", lambda M: M.ResidualBlock(128, 128), (1, 128, 32, 32), "normal"),
    Unit("down4s2_bn_lrelu_128_256", "vq", lambda M: M.Encoder(3, CH, 64).encoder[8:11], (1, 128, 32, 64), "normal"),
    Unit("res256", "vq", lambda M: M.ResidualBlock(256, 256), (1, 256, 16, 32), "normal"),
    Unit("res512", "vq", lambda M: M.ResidualBlock(512, 512), (1, 512, 8, 32), "normal"),
    Unit("conv1x1_512_64", "vq", lambda M: M.Encoder(3, CH, 64).encoder[16:17], (1, 512, 8, 32), "normal"),
    Unit("conv3_bn_lrelu_64_512", "vq", lambda M: M.Decoder(64, CH[::-1], 3).decoder[0:3], (1, 64, 8, 32), "normal"),
    Unit("up4s2_bn_lrelu_512_256", "vq", lambda M: M.Decoder(64, CH[::-1], 3).decoder[4:7], (1, 512, 4, 16), "normal"),
    Unit("up4s2_bn_lrelu_128_64", "vq", lambda M: M.Decoder(64, CH[::-1], 3).decoder[12:15], (2, 128, 16, 16), "normal"),
    Unit("conv3_sigmoid_64_3", "vq", lambda M: M.Decoder(64, CH[::-1], 3).decoder[16:18], (2, 64, 32, 32), "normal"),
    Unit("disc_conv4s2_lrelu_3_64", "vq", lambda M: M.Discriminator(3, CH).model[0:2], (2, 3, 64, 64), "image"),
    Unit("disc_conv4s2_bn_lrelu_256_512", "vq", lambda M: M.Discriminator(3, CH).model[8:11], (1, 256, 16, 64), "normal"),
    Unit("disc_conv4s1_512_1", "vq", lambda M: M.Discriminator(3, CH).model[11:12], (1, 512, 8, 32), "normal"),
    Unit("res_skipconv_32_64", "vq", lambda M: M.ResidualBlock(32, 64), (1, 32, 32, 32), "normal"),
    Unit("hier_conv3_bn_lrelu_320_128", "hier", lambda M: M.HierarchicalDecoder(3, 64, [32, 64, 128], 3).decoders[1][0][0:3],
         (1, 320, 16, 32), "normal"),
    Unit("hier_disc_conv4s1_bn_lrelu_256_512", "hier", lambda M: M.Discriminator(3, 64, 3).model[10:13], (1, 256, 9, 33), "normal"),
    Unit("hier_attention_block_128", "hier", lambda M: M.SelfAttention(128), (1, 128, 16, 32), "normal"),
]

SAMPLE = 16384     # elements of y / dx kept in the fixture (a seeded random subset) next to the whole tensor's digest


def unit_input(u, index):
    """(x, seed-derived) input of unit ``u`` and nothing else; the upstream gradient is ``cotangent('unit:'+name, y.shape)``."""
    from vae_gan_amd.testing import synthetic_eps, synthetic_images
    if u.kind == "image":
        import torch
        g = torch.Generator().manual_seed(9000 + index)
        return torch.rand(*u.shape, generator=g)
    return synthetic_eps(u.shape, 9000 + index)


def sample_index(name, numel):
    import zlib
    import torch
    if numel <= SAMPLE:
        return torch.arange(numel)
    g = torch.Generator().manual_seed(zlib.crc32(("sample:" + name).encode()) & 0x7FFFFFFF)
    return torch.randperm(numel, generator=g)[:SAMPLE]

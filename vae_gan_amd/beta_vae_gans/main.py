"""``python -m vae_gan_amd.beta_vae_gans.main [flags]`` — the flag surface of beta_vae_gans/main.py:270-336 in front of the HIP
training engine (flags, additions and the torchrun launch are described in ``vae_gan_amd/cli.py``)."""
from ..cli import build_parser as _build_parser, main as _main


def build_parser():
    return _build_parser("beta")


def main(argv=None):
    return _main("beta", argv)


if __name__ == "__main__":
    main()

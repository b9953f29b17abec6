"""``python -m vae_gan_amd.hierarchical_vae_gan.main [flags]`` — the flag surface of hierarchical_vae_gan/main.py:280-347 in front of the HIP
training engine (flags, additions and the torchrun launch are described in ``vae_gan_amd/cli.py``)."""
from ..cli import build_parser as _build_parser, main as _main


def build_parser():
    return _build_parser("hier")


def main(argv=None):
    return _main("hier", argv)


if __name__ == "__main__":
    main()

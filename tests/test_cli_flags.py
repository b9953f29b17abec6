"""CPU tier: the main.py flag surface (vae_gan_amd/cli.py) accepts every flag of the reference's three argparse
blocks with the reference's types and defaults (beta_vae_gans/main.py:273-289, vq_vae_gan/main.py:296-314,
hierarchical_vae_gan/main.py:283-299), maps them onto Config like the reference's override block, and the
engine's own additions default to the reference's behaviour."""
import pytest

from vae_gan_amd import cli

# flag -> (type, default) as written in the reference's parser.add_argument calls
COMMON = {"data_dir": (str, "./data"), "batch_size": (int, 16), "epochs": (int, 100), "lambda_adv": (float, 0.1),
          "resume": (bool, False)}
REFERENCE_FLAGS = {
    "beta": dict(COMMON, output_dir=(str, "./output/beta_vae_gan"), beta=(float, 1.0), latent_dim=(int, 128)),
    "vq": dict(COMMON, output_dir=(str, "./output/vq_vae_gan"), embedding_dim=(int, 64), num_embeddings=(int, 512),
               commitment_cost=(float, 0.25)),
    "hier": dict(COMMON, output_dir=(str, "./output/hierarchical_vae_gan"), num_hierarchies=(int, 3), beta=(float, 1.0)),
}
SAMPLE = {str: "some/dir", int: 7, float: 0.375}


@pytest.mark.parametrize("arch", cli.ARCHS)
def test_defaults_are_the_references(arch):
    args = cli.build_parser(arch).parse_args([])
    for name, (typ, default) in REFERENCE_FLAGS[arch].items():
        assert getattr(args, name) == default and isinstance(getattr(args, name), typ), name
    # additions: defaults keep the reference's behaviour (errors where the reference errors, per-rank statistics)
    assert args.bn_sync == "local" and args.recon_target_policy == "error" and args.dtype == "bf16"
    if arch == "hier":
        assert args.hier_decoder_policy == "error"


@pytest.mark.parametrize("arch", cli.ARCHS)
def test_every_reference_flag_parses_and_lands_in_config(arch):
    argv, want = [], {}
    for name, (typ, _) in REFERENCE_FLAGS[arch].items():
        if typ is bool:
            argv.append("--" + name)
            continue
        argv += ["--" + name, str(SAMPLE[typ])]
        want[name] = SAMPLE[typ]
    args = cli.build_parser(arch).parse_args(argv)
    assert args.resume is True
    cfg = cli.config_from_args(arch, args)
    for name, v in want.items():
        assert getattr(cfg, name) == v, name
    # config-only knobs keep the reference's defaults
    assert cfg.image_size == 256 and cfg.lr == 1e-4 and cfg.beta1 == 0.5 and cfg.beta2 == 0.999 and cfg.seed == 42
    assert cfg.save_interval == 5 and cfg.log_interval == 100 and cfg.device == "cuda"


def test_flags_of_other_architectures_are_rejected():
    with pytest.raises(SystemExit):
        cli.build_parser("beta").parse_args(["--num_embeddings", "512"])
    with pytest.raises(SystemExit):
        cli.build_parser("vq").parse_args(["--latent_dim", "64"])
    with pytest.raises(SystemExit):
        cli.build_parser("vq").parse_args(["--hier_decoder_policy", "nearest"])


def test_engine_additions():
    a = cli.build_parser("hier").parse_args(["--bn_sync", "global", "--recon_target_policy", "area",
                                             "--hier_decoder_policy", "nearest", "--dtype", "f32", "--image_size", "128"])
    cfg = cli.config_from_args("hier", a)
    assert (a.bn_sync, a.dtype) == ("global", "f32")
    assert (cfg.recon_target_policy, cfg.hier_decoder_policy, cfg.image_size) == ("area", "nearest", 128)
    with pytest.raises(SystemExit):
        cli.build_parser("vq").parse_args(["--bn_sync", "sometimes"])


@pytest.mark.parametrize("arch,pkg", [("vq", "vq_vae_gan"), ("beta", "beta_vae_gans"), ("hier", "hierarchical_vae_gan")])
def test_main_modules_exist_with_the_reference_layout(arch, pkg):
    import importlib
    m = importlib.import_module(f"vae_gan_amd.{pkg}.main")
    assert callable(m.main) and m.build_parser().description == cli.build_parser(arch).description

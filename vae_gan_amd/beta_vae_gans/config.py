"""Configuration surface of beta_vae_gans/config.py (same attribute names and defaults)."""


class Config:
    dataset = 'kodak'
    image_size = 256
    batch_size = 16
    num_workers = 4
    data_dir = './data'

    latent_dim = 128
    channels = 3
    enc_channels = [64, 128, 256, 512]
    dec_channels = [512, 256, 128, 64]

    lr = 1e-4
    beta1 = 0.5
    beta2 = 0.999
    epochs = 100
    save_interval = 5
    beta = 1.0
    lambda_adv = 0.1

    eval_batch_size = 1

    output_dir = './output/beta_vae_gan'
    log_interval = 100
    tensorboard_log_dir = './runs/beta_vae_gan'

    device = 'cuda'
    seed = 42

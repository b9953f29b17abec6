"""Configuration surface of hierarchical_vae_gan/config.py (same attribute names and defaults)."""


class Config:
    dataset = 'kodak'
    image_size = 256
    batch_size = 16
    num_workers = 4
    data_dir = './data'

    num_hierarchies = 3
    latent_dims = [32, 64, 128]
    base_channels = 64
    channels = 3

    lr = 1e-4
    beta1 = 0.5
    beta2 = 0.999
    epochs = 100
    save_interval = 5
    beta = 1.0
    level_weights = [1.0, 0.75, 0.5]
    lambda_adv = 0.1

    eval_batch_size = 1

    output_dir = './output/hierarchical_vae_gan'
    log_interval = 100
    tensorboard_log_dir = './runs/hierarchical_vae_gan'

    device = 'cuda'
    seed = 42

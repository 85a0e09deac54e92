"""Hyper-parameters of the path, mirroring the reference's module constants
(configs/v2_93.py:3-49, hyper_parameters.py:38; slim resnet_arg_scope defaults).
H, W and batch are runtime values here; everything else keeps the
reference's name and value."""
from dataclasses import dataclass


@dataclass
class Config:
    height: int = 288
    width: int = 512
    batch_size: int = 10
    initial_learning_rate: float = 2e-5
    feature_mul: float = 1
    theta_mul: float = 400 / 2500
    regu_mul: float = 30 / 2500
    img_mul: float = 50
    temp_mul: float = 500
    black_mul: float = 300000 / 2500
    id_mul: float = 10 / 2500
    training_iter: int = 100000
    step_size: int = 40000
    before_ch: int = 6
    after_ch: int = 0
    tot_ch: int = 7
    disp_freq: int = 100
    test_freq: int = 500
    save_freq: int = 5000
    no_theta_iter: int = 1000000
    do_temp_loss_iter: int = 5000
    do_theta_10_iter: int = -1
    do_black_loss_iter: int = 1000
    do_theta_only_iter: int = 100
    max_matches: int = 3000
    input_mask: bool = True
    do_crop_rate: float = 0.8
    indices: tuple = (0, 1, 2, 4, 8, 16, 32)
    distortion_mul: float = 1
    consistency_mul: float = 20
    grid_h: int = 4
    grid_w: int = 4
    grid_theta_mul: float = 0
    random_crop_rate: float = 0.9
    max_crop_rate: float = 0.6
    rand_H_max: tuple = ((1.1, 0.1, 0.5), (0.1, 1.1, 0.5), (0.1, 0.1, 1.0))
    rand_H_min: tuple = ((0.9, -0.1, -0.5), (-0.1, 0.9, -0.5), (-0.1, -0.1, 1.0))
    rand_H_change_rate: float = 1
    weight_decay_fc: float = 0.0002
    weight_decay_conv: float = 0.0001
    bn_eps: float = 1e-5
    bn_decay: float = 0.997
    model_dir: str = "models/v2_93/"
    log_dir: str = "log/v2_93/"

    @property
    def in_ch(self) -> int:
        return self.tot_ch + self.before_ch if self.input_mask else self.tot_ch

    @property
    def n_theta(self) -> int:
        return (self.grid_h + 1) * (self.grid_w + 1) * 2


v2_93 = Config()

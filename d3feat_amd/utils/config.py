"""Configuration class, file-compatible with the reference's `parameters.txt`.

Mirrors utils/config.py:21-170 (attribute names and defaults that the inference path and the shipped
parameters.txt files use) and :180-219 (`load`: every value is parsed with the type of the attribute's default;
`architecture`, `lr_decay_epochs`, `augment_symmetries`, `num_classes` are special-cased).  `save` writes the same
`name = value` layout (:221-313) so a round trip through either implementation is lossless for these fields.
"""
from os.path import join


class Config:
    # ---- input parameters (utils/config.py:29-53)
    gpu_id = 0
    keypts_num = 16
    det_loss_weight = 0
    safe_radius = 0.10
    dataset = ''
    network_model = ''
    num_classes = 0
    in_points_dim = 3
    in_features_dim = 1
    in_radius = 1.0
    input_threads = 8

    # ---- model parameters (:59-69)
    architecture = []
    first_features_dim = 64
    use_batch_norm = True
    batch_norm_momentum = 0.99
    segmentation_ratio = 1.0

    # ---- KPConv parameters (:75-103)
    first_subsampling_dl = 0.02
    first_kernel_radius = 0.1
    num_kernel_points = 15
    density_parameter = 3.0
    KP_extent = 1.0
    KP_influence = 'gaussian'
    convolution_mode = 'closest'
    fixed_kernel_points = 'center'
    trainable_positions = False
    modulated = False

    # ---- training parameters: parsed and kept so that parameters.txt round-trips; unused at inference (:109-166)
    learning_rate = 1e-4
    momentum = 0.9
    lr_decays = {200: 0.2, 300: 0.2}
    grad_clip_norm = 100.0
    augment_scale_anisotropic = True
    augment_scale_min = 0.9
    augment_scale_max = 1.1
    augment_symmetries = [False, False, False]
    augment_rotation = 'vertical'
    augment_noise = 0.005
    augment_occlusion = 'planar'
    augment_occlusion_ratio = 0.2
    augment_occlusion_num = 1
    augment_color = 0.7
    augment_shift_range = 0
    weights_decay = 1e-6
    gaussian_decay = 1e-3
    batch_averaged_loss = False
    points_loss = ''
    points_decay = 1e-2
    offsets_loss = 'permissive'
    offsets_decay = 1e-2
    batch_num = 10
    max_epoch = 1000
    epoch_steps = 1000
    validation_size = 100
    snapshot_gap = 50
    saving = True
    saving_path = None

    def __init__(self):
        # number of layers = number of strided / pooling blocks + 1   (utils/config.py:172-178)
        self.num_layers = len([b for b in self.architecture if 'pool' in b or 'strided' in b]) + 1

    def load(self, path):
        """utils/config.py:180-219."""
        with open(join(path, 'parameters.txt'), 'r') as f:
            lines = f.readlines()
        for line in lines:
            info = line.split()
            if len(info) > 1 and info[0] != '#':
                name = info[0]
                if info[2] == 'None':
                    setattr(self, name, None)
                elif name == 'lr_decay_epochs':
                    self.lr_decays = {int(b.split(':')[0]): float(b.split(':')[1]) for b in info[2:]}
                elif name == 'architecture':
                    self.architecture = [b for b in info[2:]]
                elif name == 'augment_symmetries':
                    self.augment_symmetries = [bool(int(b)) for b in info[2:]]
                elif name == 'num_classes':
                    self.num_classes = [int(c) for c in info[2:]] if len(info) > 3 else int(info[2])
                else:
                    attr_type = type(getattr(self, name))
                    if attr_type == bool:
                        setattr(self, name, attr_type(int(info[2])))
                    else:
                        setattr(self, name, attr_type(info[2]))
        self.saving = True
        self.saving_path = path
        self.__init__()

    def save(self, path):
        """Same `name = value` text layout as utils/config.py:221-313 (sections and value formats)."""
        def b(v):
            return '{:d}'.format(int(v))
        with open(join(path, 'parameters.txt'), 'w') as f:
            w = f.write
            w('# -----------------------------------#\n# Parameters of the training session #\n# -----------------------------------#\n\n')
            w('# Input parameters\n# ****************\n\n')
            w('dataset = {:s}\n'.format(self.dataset))
            w('network_model = {:s}\n'.format(self.network_model))
            if isinstance(self.num_classes, list):
                w('num_classes =' + ''.join(' {:d}'.format(n) for n in self.num_classes) + '\n')
            else:
                w('num_classes = {:d}\n'.format(self.num_classes))
            w('in_points_dim = {:d}\n'.format(self.in_points_dim))
            w('in_features_dim = {:d}\n'.format(self.in_features_dim))
            w('in_radius = {:.3f}\n'.format(self.in_radius))
            w('input_threads = {:d}\n\n'.format(self.input_threads))
            w('# Model parameters\n# ****************\n\n')
            w('architecture =' + ''.join(' {:s}'.format(a) for a in self.architecture) + '\n')
            w('num_layers = {:d}\n'.format(self.num_layers))
            w('first_features_dim = {:d}\n'.format(self.first_features_dim))
            w('use_batch_norm = ' + b(self.use_batch_norm) + '\n')
            w('batch_norm_momentum = {:.3f}\n\n'.format(self.batch_norm_momentum))
            w('segmentation_ratio = {:.3f}\n\n'.format(self.segmentation_ratio))
            w('# KPConv parameters\n# *****************\n\n')
            w('first_subsampling_dl = {:.3f}\n'.format(self.first_subsampling_dl))
            w('num_kernel_points = {:d}\n'.format(self.num_kernel_points))
            w('density_parameter = {:.3f}\n'.format(self.density_parameter))
            w('fixed_kernel_points = {:s}\n'.format(self.fixed_kernel_points))
            w('KP_extent = {:.3f}\n'.format(self.KP_extent))
            w('KP_influence = {:s}\n'.format(self.KP_influence))
            w('convolution_mode = {:s}\n'.format(self.convolution_mode))
            w('trainable_positions = ' + b(self.trainable_positions) + '\n\n')
            w('modulated = ' + b(self.modulated) + '\n\n')
            w('# Training parameters\n# *******************\n\n')
            w('learning_rate = {:f}\n'.format(self.learning_rate))
            w('momentum = {:f}\n'.format(self.momentum))
            w('lr_decay_epochs =' + ''.join(' {:d}:{:f}'.format(e, d) for e, d in self.lr_decays.items()) + '\n')
            w('grad_clip_norm = {:f}\n\n'.format(self.grad_clip_norm))
            w('augment_symmetries =' + ''.join(' ' + b(a) for a in self.augment_symmetries) + '\n')
            w('augment_rotation = {:s}\n'.format(str(self.augment_rotation)))
            w('augment_noise = {:f}\n'.format(self.augment_noise))
            w('augment_occlusion = {:s}\n'.format(str(self.augment_occlusion)))
            w('augment_occlusion_ratio = {:.3f}\n'.format(self.augment_occlusion_ratio))
            w('augment_occlusion_num = {:d}\n'.format(self.augment_occlusion_num))
            w('augment_scale_anisotropic = ' + b(self.augment_scale_anisotropic) + '\n')
            w('augment_scale_min = {:.3f}\n'.format(self.augment_scale_min))
            w('augment_scale_max = {:.3f}\n'.format(self.augment_scale_max))
            w('augment_color = {:.3f}\n\n'.format(self.augment_color))
            w('weights_decay = {:f}\n'.format(self.weights_decay))
            w('gaussian_decay = {:f}\n'.format(self.gaussian_decay))
            w('batch_averaged_loss = ' + b(self.batch_averaged_loss) + '\n')
            w('offsets_loss = {:s}\n'.format(self.offsets_loss))
            w('offsets_decay = {:f}\n'.format(self.offsets_decay))
            w('batch_num = {:d}\n'.format(self.batch_num))
            w('max_epoch = {:d}\n'.format(self.max_epoch))
            w('epoch_steps = {:d}\n'.format(self.epoch_steps))
            w('validation_size = {:d}\n'.format(self.validation_size))
            w('snapshot_gap = {:d}\n'.format(self.snapshot_gap))


def threedmatch_config():
    """The shipped 3DMatch model: results/Log_contraloss/parameters.txt (identical to training_3DMatch.py:40-98)."""
    c = Config()
    c.dataset = '3DMatch'
    c.network_model = 'descriptor'
    c.architecture = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                      'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb', 'nearest_upsample', 'unary',
                      'nearest_upsample', 'unary', 'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
                      'last_unary']
    c.first_features_dim = 64
    c.batch_norm_momentum = 0.98
    c.first_subsampling_dl = 0.03
    c.num_kernel_points = 15
    c.density_parameter = 5.0
    c.KP_extent = 1.0
    c.KP_influence = 'linear'
    c.convolution_mode = 'sum'
    c.batch_num = 1
    c.__init__()
    return c


def kitti_config():
    """results_kitti/Log_11011605/parameters.txt: same network, first_subsampling_dl = 0.30."""
    c = threedmatch_config()
    c.dataset = 'KITTI'
    c.first_subsampling_dl = 0.30
    return c

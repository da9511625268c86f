"""The HOCON-subset reader (nefii_amd/conf.py) that stands in for pyhocon: the constructs the reference's conf files use,
and - in the build container, where the files exist - the four files of code/confs_sg themselves."""
import os

import pytest

from nefii_amd import conf

REF_CONFS = '/root/reference/code/confs_sg'

TEXT = '''
# comment line
train{
    expname = default          # trailing comment
    dataset_class = datasets.scene_dataset.SceneDataset
    num_pixels = 2048
    idr_learning_rate = 5e-4
    alpha_milestones = [250000,500000, 750000]   // another comment style
    alpha_factor = 2
    empty = []
}
loss { r_patch = 1.0
       loss_type = L1 }
model = {
    render_background = True
    fast_multi_ray = False
    implicit_network
    {
        dims = [ 512, 512,
                 512 ]
        bias = 0.6
        use_last_as_f=True
        name = "quoted string # not a comment"
    }
    nested { deeper { value = -3 } }
}
'''


def test_parser_constructs():
    c = conf.parse_string(TEXT)
    assert isinstance(c, conf.ConfigTree)
    assert c.get_string('train.expname') == 'default'
    assert c.get_string('train.dataset_class') == 'datasets.scene_dataset.SceneDataset'
    assert c.get_int('train.num_pixels') == 2048 and isinstance(c.get_int('train.num_pixels'), int)
    assert c.get_float('train.idr_learning_rate') == 5e-4
    assert c.get_list('train.alpha_milestones') == [250000, 500000, 750000]
    assert c.get_list('train.empty') == [] and c.get_int('train.alpha_factor') == 2
    assert c.get_float('loss.r_patch') == 1.0 and c.get_string('loss.loss_type') == 'L1'
    assert c.get_bool('model.render_background') is True and c.get_bool('model.fast_multi_ray') is False
    net = c.get_config('model.implicit_network')
    assert net.get_list('dims') == [512, 512, 512] and net.get_bool('use_last_as_f') is True
    assert net.get_float('bias') == 0.6 and net.get_string('name') == 'quoted string # not a comment'
    assert c.get_int('model.nested.deeper.value') == -3
    assert dict(**c.get_config('loss')) == {'r_patch': 1.0, 'loss_type': 'L1'}          # the **kwargs use
    # defaults and missing keys follow pyhocon: default returned when given, KeyError otherwise
    assert c.get_int('train.num_rays', default=-1) == -1 and c.get_list('train.nope', default=[]) == []
    assert c.get_float('train.idr_sched_factor', default=0.0) == 0.0
    with pytest.raises(KeyError):
        c.get_int('train.nope')
    d = conf.from_dict({'a': {'b': 1}, 'c': [1, 2]})
    assert d.get_int('a.b') == 1 and d.get_config('a').get_int('b') == 1 and d.get_list('c') == [1, 2]


@pytest.mark.skipif(not os.path.isdir(REF_CONFS), reason='reference conf files exist in the build container only')
def test_reference_conf_files_parse():
    want = {
        'conf.conf': {'train.num_pixels': 2048, 'train.num_rays': 64, 'model.render_type': 'pt_render_indirect_mlp',
                      'model.feature_vector_size': 512, 'model.implicit_network.multires': 6,
                      'model.envmap_material_network.num_lgt_sgs': 128, 'loss.mask_weight': 100.0},
        'physg.conf': {'model.implicit_network.skip_in': [4]},
        'conf_neus.conf': {'model.implicit_network.dims': [256] * 8},
        'sdf.conf': {'train.model_class': 'model.implicit_differentiable_renderer.IDRNetwork'},
    }
    for name, checks in want.items():
        c = conf.parse_file(os.path.join(REF_CONFS, name))
        for sect in ('train', 'loss', 'model'):
            assert isinstance(c.get_config(sect), conf.ConfigTree), (name, sect)
        for key, val in checks.items():
            got = c.get(key)
            assert got == val, (name, key, got)
        m = c.get_config('model')
        assert m.get_config('implicit_network').get_bool('weight_norm') is True
        assert len(m.get_config('rendering_network').get_list('dims')) >= 1
        from nefii_amd.utils import general
        assert general.get_class(c.get_string('train.model_class')).__name__ == 'IDRNetwork'
        assert general.get_class(c.get_string('train.dataset_class')).__name__ == 'SceneDataset'


def test_tracer_bisection_levels_follow_the_batch_size():
    """RayTracing.auto_levels: 5 speculative bisection levels for a lone latency-bound batch, 3 for mid-size batches and
    for traces that run beside other work, none (1 level per round) once the batch fills the chip on its own."""
    from nefii_amd.model.ray_tracing import RayTracing
    assert RayTracing.auto_levels(512) == 5 and RayTracing.auto_levels(16384) == 5
    assert RayTracing.auto_levels(4096, concurrent=True) == 3 and RayTracing.auto_levels(65536) == 3
    assert RayTracing.auto_levels(131072) == 1 and RayTracing.auto_levels(786432, concurrent=True) == 1


def test_tensorboard_event_file_round_trip(tmp_path):
    """utils/tb_writer.py: the event file the runner writes where the reference's tensorboardX SummaryWriter would
    (idr_train.py:114-115,881-895) - TFRecord framing with masked CRC32C (known vector: '123456789' -> 0xE3069283), Event /
    Summary messages for scalars and PNG images - read back record by record, checksums verified, values exact."""
    import io
    import numpy as np
    import torch
    from PIL import Image
    from nefii_amd.utils import tb_writer as tb
    assert tb.crc32c(b'123456789') == 0xE3069283
    w = tb.SummaryWriter(str(tmp_path))
    w.add_scalar('sg_psnr', 31.25, 50)
    w.add_scalar('loss', torch.tensor(0.125), 100)
    img = np.zeros((3, 4, 6), np.float32)
    img[0, 1, 2] = 1.0
    img[2] = 0.5
    w.add_image('train/panel-0', torch.from_numpy(img), 100)
    w.close()
    files = [f for f in os.listdir(str(tmp_path)) if f.startswith('events.out.tfevents.')]
    assert len(files) == 1
    ev = tb.read_events(os.path.join(str(tmp_path), files[0]))
    assert ev[0]['file_version'] == 'brain.Event:2' and len(ev) == 4
    assert ev[1]['tag'] == 'sg_psnr' and ev[1]['value'] == 31.25 and ev[1]['step'] == 50
    assert ev[2]['tag'] == 'loss' and ev[2]['value'] == 0.125 and ev[2]['step'] == 100
    h, wd, c, png = ev[3]['image']
    assert (h, wd, c) == (4, 6, 3) and ev[3]['tag'] == 'train/panel-0'
    back = np.asarray(Image.open(io.BytesIO(png)))
    assert back.shape == (4, 6, 3) and back[1, 2, 0] == 255 and back[0, 0, 2] == 128 and back[0, 0, 0] == 0


def test_scheduler_fast_path_equals_multisteplr():
    """TrainStep steps its two MultiStepLR schedulers through _scheduler_step: off the milestones two counter increments instead of
    torch's step() (which rewrites and clones an unchanged learning rate - launches, in graph mode).  Learning rates, get_last_lr
    and the state dict (what the checkpoints hold, idr_train.py:251-306) must be what step() leaves, repeated milestones included."""
    import torch
    from nefii_amd.training.step import _scheduler_step

    def run(fast):
        p = [torch.nn.Parameter(torch.zeros(3))]
        o = torch.optim.Adam(p, lr=5e-4)
        s = torch.optim.lr_scheduler.MultiStepLR(o, [3, 7, 7, 12], gamma=0.5)
        seen = []
        for _ in range(15):
            o.step()
            _scheduler_step(s) if fast else s.step()
            seen.append((o.param_groups[0]['lr'], tuple(s.get_last_lr()), s.last_epoch, s._step_count))
        return seen, dict(s.state_dict())

    a, sa = run(True)
    b, sb = run(False)
    assert a == b
    assert sa == sb

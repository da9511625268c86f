"""Multi-process CPU tests (gloo, world_size 2) of what the path does across GPUs: the per-rank pixel shard, the
single flat gradient all-reduce (mean) of a training step, and the render chunk / round-robin / gather / merge
contract.  No GPU and no HIP compute involved: a small torch module stands in for the renderer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nefii_amd import synthetic as syn
from nefii_amd.training import render as R
from nefii_amd.training.step import allreduce_mean_gradients
from nefii_amd.utils import general as utils


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeRenderer(torch.nn.Module):
    """Per-pixel deterministic function of uv with the output keys of IDRNetwork.forward."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.lin = torch.nn.Linear(2, 3)
        with torch.no_grad():
            self.lin.weight.copy_(torch.randn(3, 2, generator=g) * 0.01)
            self.lin.bias.copy_(torch.randn(3, generator=g))

    def forward(self, inp):
        uv = inp['uv'].reshape(-1, 2)
        rgb = torch.sigmoid(self.lin(uv))
        n = uv.shape[0]
        out = {k: rgb * (i + 1) for i, (k, w) in enumerate(R.RENDER_KEYS) if w == 3}
        out['sg_roughness_values'] = rgb[:, :1]
        out['network_object_mask'] = rgb[:, 0] > 0.5
        out['object_mask'] = inp['object_mask'].reshape(-1)
        assert n == out['object_mask'].shape[0]
        return out


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    # ---- training: shard + one flat all-reduce == single-process gradient of the mean of per-rank losses
    model = FakeRenderer()
    inp, gt = syn.make_inputs(64, (32, 32), 40.0, (0., 0., 3.), -1, seed=3, rank=rank, world_size=world)
    out = model(inp)
    loss = (out['sg_rgb_values'] - gt.reshape(-1, 3)).abs().mean()
    loss.backward()
    nbytes, _ = allreduce_mean_gradients(list(model.parameters()), world)
    assert nbytes == sum(p.numel() for p in model.parameters()) * 4
    torch.save([p.grad.clone() for p in model.parameters()], os.path.join(tmp, 'grad%d.pt' % rank))
    # ---- rendering: chunk / round-robin / gather / merge
    H = W = 12
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    uv = torch.stack([xs, ys], -1).reshape(1, -1, 2).float()
    full = {'uv': uv, 'object_mask': torch.ones(1, H * W, dtype=torch.bool), 'pose': torch.eye(4)[None],
            'intrinsics': torch.eye(4)[None]}
    model.eval()
    merged = R.render_frame(model, full, H * W, num_rays=1, memory_capacity_level=5, rank=rank, world_size=world)
    if rank == 0:
        torch.save(merged, os.path.join(tmp, 'render.pt'))
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_shards_allreduce_and_render_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    # reference result in one process
    model = FakeRenderer()
    grads = []
    uvs = []
    for r in range(world):
        model.zero_grad()
        inp, gt = syn.make_inputs(64, (32, 32), 40.0, (0., 0., 3.), -1, seed=3, rank=r, world_size=world)
        uvs.append(inp['uv'])
        loss = (model(inp)['sg_rgb_values'] - gt.reshape(-1, 3)).abs().mean()
        loss.backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    mean = [(a + b) / 2 for a, b in zip(*grads)]
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, 'grad%d.pt' % r))
        for g, m in zip(got, mean):
            assert torch.allclose(g, m, atol=1e-7)
    # the per-rank shards partition the single-process batch, contiguously (scene_dataset.py:268-279)
    whole, _ = syn.make_inputs(64, (32, 32), 40.0, (0., 0., 3.), -1, seed=3)
    assert torch.equal(torch.cat(uvs, dim=1), whole['uv'])
    # render: merged multi-rank frame == single-process frame
    H = W = 12
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    uv = torch.stack([xs, ys], -1).reshape(1, -1, 2).float()
    full = {'uv': uv, 'object_mask': torch.ones(1, H * W, dtype=torch.bool), 'pose': torch.eye(4)[None],
            'intrinsics': torch.eye(4)[None]}
    model.eval()
    single = R.render_frame(model, full, H * W, num_rays=1, memory_capacity_level=5)
    multi = torch.load(os.path.join(tmp_path, 'render.pt'))
    assert set(single.keys()) == set(multi.keys())
    for k in single:
        # chunk sizes differ (level - log2 W), so vectorised CPU math may differ in the last bit
        if single[k].dtype == torch.bool:
            assert torch.equal(single[k], multi[k]), k
        else:
            assert torch.allclose(single[k], multi[k], atol=1e-6), k
    direct = model(full)
    assert torch.allclose(single['sg_rgb_values'], direct['sg_rgb_values'])


@pytest.mark.parametrize('n,world', [(10, 1), (10, 2), (11, 4), (5000, 8), (3, 4)])
def test_chunk_plan_assigns_every_chunk_once(n, world):
    order, slices = R.plan_chunks(n, world)
    assert sorted(order) == list(range(n))
    covered = []
    for r, (a, b) in enumerate(slices):
        covered += list(range(a, b))
        assert utils.scatter_list(order, n, r, world) == order[a:b]
    assert covered == list(range(n))


def test_split_and_merge_roundtrip():
    uv = torch.arange(2 * 37 * 2, dtype=torch.float32).reshape(2, 37, 2)
    inp = {'uv': uv, 'object_mask': torch.ones(2, 37, dtype=torch.bool)}
    split = utils.split_input(inp, 37, num_rays=4, memory_capacity_level=5)      # 8 pixels per chunk
    assert [s['uv'].shape[1] for s in split] == [8, 8, 8, 8, 5]
    res = [{'a': s['uv'].reshape(-1, 2), 'm': s['object_mask'].reshape(-1)} for s in split]
    merged = utils.merge_output(res, 37, 2)
    assert torch.equal(merged['a'], uv.reshape(-1, 2))
    assert merged['m'].shape == (74,)


def test_synthetic_dataset_follows_scene_dataset_contract():
    """scene_dataset.py:149-279: item layout, 2x2 patch sampling, contiguous per-rank split of the patch list (the last
    rank takes the remainder), sub-pixel ray jitter shared by all pixels, collate."""
    import numpy as np
    import torch
    from nefii_amd.datasets.synthetic_dataset import SyntheticSceneDataset
    ds = SyntheticSceneDataset(n_views=3, img_res=(20, 24))
    idx, sample, gt = ds[1]
    assert sample['uv'].shape == (480, 2) and gt['rgb'].shape == (480, 3) and sample['pose'].shape == (4, 4)
    assert sample['uv'][25].tolist() == [1.0, 1.0]                      # (u = x, v = y), row-major pixels
    np.random.seed(3)
    ds.change_sampling_idx_patch(10, 1)
    full = ds.sampling_idx.clone()
    assert full.shape == (40,)
    p = full.reshape(10, 4)
    assert ((p[:, 1] - p[:, 0]) == 1).all() and ((p[:, 2] - p[:, 0]) == 24).all() and ((p[:, 3] - p[:, 2]) == 1).all()
    parts = []
    for rank in range(3):
        ds.sampling_idx = full.clone()
        ds.scatter_sampling_idx_patch(rank, 3, 10, 1)
        parts.append(ds.sampling_idx)
    assert [t.shape[0] for t in parts] == [12, 12, 16]
    assert torch.equal(torch.cat(parts), full)
    ds.change_sampling_rays(5)
    idx, sample, gt = ds[0]
    assert sample['uv'].shape == (16, 5, 2) and gt['rgb'].shape == (16, 3) and sample['object_mask'].shape == (16,)
    jit = sample['uv'] - sample['uv'].round()
    assert (jit[0] - jit[7]).abs().max() < 1e-6                          # one jitter set for every pixel
    batch = ds.collate_fn([ds[0], ds[2]])
    assert batch[0].tolist() == [0, 2] and batch[1]['uv'].shape == (2, 16, 5, 2) and batch[2]['rgb'].shape == (2, 16, 3)


def test_chunking_helpers_match_the_reference():
    """split_input / merge_output / scatter_list and render.py's round-robin chunk order, against arrays produced by the
    reference's own functions (tests/golden/make_general_golden.py)."""
    import os
    import numpy as np
    from nefii_amd.training import render as R
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'general_ref.npz')))
    for tag, total, num_rays, level, batch in (('single', 77, -1, 4, 1), ('multi', 50, 3, 5, 2)):
        uv, mask = torch.from_numpy(g[tag + '_uv']), torch.from_numpy(g[tag + '_mask'])
        inp = {'uv': uv, 'object_mask': mask, 'pose': torch.eye(4)[None].repeat(batch, 1, 1)}
        split = utils.split_input(inp, total, num_rays, level)
        assert [s['uv'].shape[1] for s in split] == g[tag + '_sizes'].tolist()
        assert np.array_equal(split[1]['uv'].numpy(), g[tag + '_first_uv'])
        assert split[0]['pose'] is inp['pose'] or torch.equal(split[0]['pose'], inp['pose'])
        res = [{'a': s['uv'].reshape(batch, s['uv'].shape[1], -1).sum(-1).reshape(-1),
                'b': s['uv'].reshape(batch, s['uv'].shape[1], -1)[..., :2].reshape(-1, 2), 'none': None} for s in split]
        merged = utils.merge_output(res, total, batch)
        assert 'none' not in merged
        assert np.array_equal(merged['a'].numpy(), g[tag + '_merged_a'])
        assert np.array_equal(merged['b'].numpy(), g[tag + '_merged_b'])
        for world in (2, 3):
            order, slices = R.plan_chunks(len(split), world)
            assert order == g['%s_order_w%d' % (tag, world)].tolist()
            for rank in range(world):
                want = g['%s_scatter_w%d_r%d' % (tag, world, rank)].tolist()
                assert order[slices[rank][0]:slices[rank][1]] == want
                assert utils.scatter_list(order, len(order), rank, world) == want


# ---- TrainStep across ranks: fixed gradient layout, collective skip decisions, initial broadcast, NaN guard ---------
class FakeIDR(torch.nn.Module):
    """Stand-in with IDRNetwork's module surface and output keys (no HIP compute): hit pixels are shaded by two small
    linear maps, pixels that miss keep the constant defaults - so a batch without a single hit carries no gradient."""

    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.implicit_network = torch.nn.Linear(3, 1)
        self.rendering_network = torch.nn.Linear(2, 3)
        self.envmap_material_network = torch.nn.Linear(2, 3)
        self.unused = torch.nn.Parameter(torch.randn(5, generator=g))       # never receives a gradient on any rank
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        for p in self.implicit_network.parameters():
            p.requires_grad = False
        self.render_type = 'pt'
        self.poison = False

    def forward(self, inp, with_point=False):
        if with_point:
            x = inp['points'][:, 0, :2]
            return {'idr_rgb_values': self.rendering_network(x), 'sg_rgb_values': self.envmap_material_network(x)}
        uv = inp['uv'].reshape(-1, 2) * 0.05
        hit = inp['hit'].reshape(-1)
        n = uv.shape[0]
        ones = torch.ones(n, 3)
        idr = torch.where(hit[:, None], torch.sigmoid(self.rendering_network(uv)), ones)
        sg = torch.where(hit[:, None], torch.sigmoid(self.envmap_material_network(uv)), ones)
        if not hit.any():
            idr, sg = ones.clone(), ones.clone()
        if self.poison:
            sg = sg * float('nan')
        sec = inp.get('sec_mask')
        return {'idr_rgb_values': idr, 'sg_rgb_values': sg, 'network_object_mask': hit,
                'object_mask': inp['object_mask'].reshape(-1), 'sdf_output': torch.full((n, 1), 0.1),
                'normal_values': ones, 'grad_theta': None,
                'secondary_points': torch.cat([uv, uv[:, :1]], 1).reshape(1, n, 3) if sec is not None else None,
                'secondary_mask': sec.reshape(1, n, 1) if sec is not None else None,
                'secondary_dir': torch.ones(1, n, 3) if sec is not None else None}


def _fake_batch(rank, hits, sec=None):
    inp, gt = syn.make_inputs(32, (32, 32), 40.0, (0., 0., 3.), -1, seed=5 + rank)
    n = inp['uv'].shape[1]
    inp['hit'] = torch.full((n,), bool(hits))
    if sec is not None:
        inp['sec_mask'] = torch.full((n,), bool(sec))
    return inp, {'rgb': gt}


LOSS_CONF = dict(idr_rgb_weight=1.0, sg_rgb_weight=1.0, eikonal_weight=0.0, mask_weight=1.0, alpha=50.0)


def _step_worker(rank, world, port, tmp):
    from nefii_amd.training.step import TrainStep
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = FakeIDR(seed=10 + rank)                 # every rank starts from its own initialisation ...
    st = TrainStep(model, LOSS_CONF, world_size=world, secondary_train_interval=2, secondary_batch_size=64, num_rays=2)
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(tmp, 'init%d.pt' % rank))
    # iteration 0: rank 1's slice has no hit at all (no gradient there) and no secondary hit (its secondary step has
    # nothing to differentiate); iteration 1: rank 1's loss is not finite (no secondary step: 1 % 2); iteration 2: everyone
    # has hits and secondary hits; iteration 4 (a secondary iteration again): rank 1 has no primary hit, so its forward
    # returns NO secondary outputs at all (IDRNetwork.shade_tail: ret = {}) while rank 0 trains on its secondary hits
    plan = [(rank == 0, rank == 0), (True, None), (True, True), (True, None), (rank == 0, True if rank == 0 else None)]
    for it, (hits, sec) in enumerate(plan):
        model.poison = it == 1 and rank == 1
        inp, gt = _fake_batch(rank, hits, sec)
        st(inp, gt)
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(tmp, 'it%d_r%d.pt' % (it, rank)))
    torch.save(st.nonfinite_steps.clone(), os.path.join(tmp, 'bad%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_collectives_do_not_depend_on_the_data(tmp_path):
    """A rank without hits (no gradients), a rank without secondary hits, and a rank whose loss is NaN all enter the same
    collectives as their peers with same-sized buffers; parameters stay identical across ranks and finite; the initial
    parameters are rank 0's (DDP's construction-time broadcast, idr_train.py:308-309)."""
    from nefii_amd.training.step import TrainStep
    world = 2
    mp.spawn(_step_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    L = lambda name: torch.load(os.path.join(tmp_path, name))
    ref0 = FakeIDR(seed=10).state_dict()
    for r in range(world):
        got = L('init%d.pt' % r)
        for k in ref0:
            assert torch.equal(got[k], ref0[k]), (r, k)
    for it in range(5):
        a, b = L('it%d_r0.pt' % it), L('it%d_r1.pt' % it)
        for k in a:
            assert torch.equal(a[k], b[k]), (it, k)
            assert torch.isfinite(a[k]).all(), (it, k)
    assert L('bad0.pt').item() == 1 and L('bad1.pt').item() == 1
    # the NaN iteration is SKIPPED on every rank, not run on zero gradients: parameters are exactly those of iteration 0
    # (Adam on a zero gradient would still have moved them by the first moment), and the next iterations train again
    a, b = L('it0_r0.pt'), L('it1_r0.pt')
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = L('it2_r0.pt')
    assert any(not torch.equal(b[k], c[k]) for k in b)
    d, e = L('it3_r0.pt'), L('it4_r0.pt')
    assert any(not torch.equal(d[k], e[k]) for k in d)
    # iteration 0 against one process: primary gradient = (rank 0's gradient + 0) / 2, then the secondary step likewise
    model = FakeIDR(seed=10)
    st = TrainStep(model, LOSS_CONF, world_size=1, secondary_train_interval=0, num_rays=2)
    inp, gt = _fake_batch(0, True, True)
    out = model(inp)
    lo = st.loss(out, gt)
    lo['loss'].backward()
    for p in model.parameters():
        if p.grad is not None:
            p.grad.mul_(0.5)
    st.idr_optimizer.step()
    st.sg_optimizer.step()
    st.world_size = 2           # halve the secondary gradient the same way: emulate by hand
    idx = torch.nonzero(out['secondary_mask'].reshape(-1)).flatten()[:32]      # secondary_batch_size // world
    p_ = out['secondary_points'].detach().reshape(-1, 3).index_select(0, idx)
    ret = model({'points': p_.unsqueeze(1).expand(-1, 2, 3), 'ray_dirs': p_.unsqueeze(1).expand(-1, 2, 3)}, with_point=True)
    st.idr_optimizer.zero_grad()
    st.sg_optimizer.zero_grad()
    torch.nn.functional.l1_loss(ret['sg_rgb_values'], ret['idr_rgb_values']).backward()
    for p in model.parameters():
        if p.grad is not None:
            p.grad.mul_(0.5)
    st.idr_optimizer.step()
    st.sg_optimizer.step()
    got = L('it0_r1.pt')
    for k, v in model.state_dict().items():
        assert torch.allclose(got[k], v, atol=1e-6), k
    assert torch.equal(got['unused'], ref0['unused'])         # no gradient on any rank: untouched


@pytest.mark.parametrize('world', [1, 2, 8])
def test_config5_frame_plan_matches_the_reference(world):
    """BASELINE config 5 at its real size (800 x 800 pixels, 256 rays per pixel, memory_capacity_level 18): the chunk
    sizes, the round-robin order and every rank's slice equal what the reference's split_input / scatter_list produce
    (tests/golden/make_general_golden.py ran them for W = 1, 2, 8)."""
    import math
    import numpy as np
    w = syn.WORKLOADS['cfg5']
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'general_ref.npz')))
    total = w['image_hw'][0] * w['image_hw'][1]
    assert total == w['num_pixels']
    level = w['memory_capacity_level'] - int(math.floor(math.log2(world)))
    uv = torch.zeros(1, total, 2)
    uv[0, :, 0] = torch.arange(total)
    split = utils.split_input({'uv': uv, 'object_mask': torch.ones(1, total, dtype=torch.bool)}, total, w['num_rays'], level)
    assert [s['uv'].shape[1] for s in split] == g['cfg5_w%d_sizes' % world].tolist()
    assert [int(s['uv'][0, 0, 0]) for s in split] == g['cfg5_w%d_first_pixel' % world].tolist()
    order, slices = R.plan_chunks(len(split), world)
    assert order == g['cfg5_w%d_order' % world].tolist()
    lens = g['cfg5_w%d_scatter_lens' % world].tolist()
    assert [b - a for a, b in slices] == lens
    flat = g['cfg5_w%d_scatter' % world].tolist()
    off = 0
    for rank, (a, b) in enumerate(slices):
        assert order[a:b] == flat[off:off + lens[rank]]
        off += lens[rank]
    # every rank renders the same number of pixels of the frame (xGMI gather of equal-sized packed buffers)
    px = [sum(split[c]['uv'].shape[1] for c in order[a:b]) for a, b in slices]
    assert len(set(px)) == 1 and sum(px) == total


def test_flat_grads_are_views_of_one_buffer():
    """training/step.py:FlatGrads - the multi-rank step's gradients: every parameter's .grad is a view into ONE flat buffer
    (so the exchange is one all-reduce of the buffer as it stands), zero() is one fill that also re-points .grad, backward
    accumulates into the views IN PLACE, adopt() takes over gradients that landed in tensors of their own, and the flag slot
    rides behind the parameters."""
    from nefii_amd.training.step import FlatGrads
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    params = list(net.parameters())
    fg = FlatGrads(params, n_flags=1)
    n_par = sum(p.numel() for p in params)
    assert fg.flat.numel() == n_par + 1 and fg.flags.data_ptr() == fg.flat[n_par:].data_ptr()
    fg.zero()
    ptrs = [p.grad.data_ptr() for p in params]
    assert all(p.grad is v for p, v in zip(params, fg.views))
    x = torch.randn(11, 5)
    net(x).square().sum().backward()
    assert [p.grad.data_ptr() for p in params] == ptrs, 'backward replaced a gradient view instead of accumulating into it'
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    assert torch.equal(fg.flat[:n_par], ref) and ref.abs().sum() > 0
    # a second backward accumulates; zero() clears through the views
    net(x).square().sum().backward()
    assert torch.allclose(fg.flat[:n_par], 2 * ref)
    fg.zero()
    assert fg.flat.abs().sum() == 0 and all(p.grad.abs().sum() == 0 for p in params)
    # gradients that ended up elsewhere (optimizer.zero_grad(set_to_none=True) + backward; a captured graph's own tensors)
    for p in params:
        p.grad = None
    net(x).square().sum().backward()
    assert all(p.grad.data_ptr() != q for p, q in zip(params, ptrs))
    fg.adopt()
    assert [p.grad.data_ptr() for p in params] == ptrs and torch.allclose(fg.flat[:n_par], ref)
    # mean over "ranks" without a process group: world_size 1 is a no-op elsewhere; here only the scale is checked
    fg.flags.fill_(1.0)
    fg.flat[:n_par].mul_(1.0 / 4)
    assert torch.allclose(fg.flat[:n_par], ref / 4) and fg.flags.item() == 1.0

"""Step-1 data side: mesh loading and signed-distance sampling, on CPU."""
import numpy as np
import pytest
import torch

from nefii_amd.datasets.sdf_dataset import MeshSDF, SDFDataset, SDFSampler, load_obj


def box_mesh(lo, hi):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    v = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))])
    return v, f, quads


def box_sdf(p, lo, hi):
    c, h = (np.asarray(lo) + np.asarray(hi)) / 2, (np.asarray(hi) - np.asarray(lo)) / 2
    q = np.abs(p - c) - h
    return np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)


def test_mesh_sdf_is_exact_on_a_box():
    """face, edge and corner regions, inside and outside, against the closed form"""
    lo, hi = (-0.3, -0.2, -0.45), (0.5, 0.35, 0.1)
    v, f, _ = box_mesh(lo, hi)
    g = np.random.Generator(np.random.Philox(2))
    p = g.uniform(-0.9, 0.9, size=(4000, 3))
    d = MeshSDF(v, f)(p).numpy()
    ref = box_sdf(p, lo, hi)
    assert np.abs(d - ref).max() < 1e-12
    assert (d < 0).sum() > 100 and (d > 0).sum() > 100
    # winding does not matter for the sign (parity), nor do degenerate faces
    f2 = np.concatenate([f[:, ::-1], [[0, 0, 1]]])
    assert np.abs(MeshSDF(v, f2)(p).numpy() - ref).max() < 1e-12
    assert np.abs(MeshSDF(v, f, pair_budget=100)(p[:50]).numpy() - ref[:50]).max() < 1e-12      # chunked evaluation


def test_load_obj_polygons_negative_and_slash_indices(tmp_path):
    lo, hi = (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)
    v, f, quads = box_mesh(lo, hi)
    lines = ['# box', 'mtllib x.mtl'] + ['v %r %r %r' % tuple(float(t) for t in x) for x in v] + ['vn 0 0 1', 'vt 0 0']
    for i, q in enumerate(quads):
        if i % 3 == 0:
            lines.append('f ' + ' '.join(str(k + 1) for k in q))
        elif i % 3 == 1:
            lines.append('f ' + ' '.join('%d/1/1' % (k + 1) for k in q))
        else:
            lines.append('f ' + ' '.join('%d//1' % (k - len(v)) for k in q))
    path = tmp_path / 'box.obj'
    path.write_text('\n'.join(lines))
    v2, f2 = load_obj(str(path))
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    (tmp_path / 'empty.obj').write_text('# nothing\n')
    with pytest.raises(ValueError):
        load_obj(str(tmp_path / 'empty.obj'))


@pytest.mark.parametrize('scale_to_unit', [True, False])
def test_sampler_distribution_and_unit_scaling(scale_to_unit):
    """47/50 of the samples hug the surface (two noise levels), the rest fill the unit ball; with scale_to_unit the mesh
    is normalised for sampling and positions / distances are mapped back (sdf_dataset.py:52-55)."""
    lo, hi = np.array([0.1, 0.0, -0.2]), np.array([0.4, 0.2, 0.0])
    v, f, _ = box_mesh(lo, hi)
    s = SDFSampler(None, 2000, scale_to_unit=scale_to_unit, mesh=(v, f))
    if scale_to_unit:
        assert np.allclose(s.center, (lo + hi) / 2) and np.isclose(s.scale, np.linalg.norm((hi - lo) / 2))
    else:
        assert np.all(s.center == 0) and s.scale == 1.0
    pts, sdf = s.sample(torch.Generator().manual_seed(5))
    assert pts.shape == (2000, 3) and sdf.shape == (2000, 1)
    assert np.abs(sdf[:, 0].numpy() - box_sdf(pts.numpy(), lo, hi)).max() < 1e-10
    near = 2000 * 47 // 50 // 2
    d = sdf[:, 0].abs() / s.scale
    assert d[:near].max() < 6 * 0.0025 and d[near:2 * near].max() < 6 * 0.00025
    assert 0.0012 < d[:near].mean() < 0.0028 and 0.00012 < d[near:2 * near].mean() < 0.00028
    ball = (pts[2 * near:] - torch.from_numpy(s.center)) / s.scale
    assert ball.shape[0] == 2000 - 2 * near and ball.norm(dim=1).max() <= 1.0 + 1e-9
    assert 0.6 < ball.norm(dim=1).mean() < 0.9                  # E|x| = 3/4 for the uniform ball


def test_sdf_dataset_items_and_collate():
    v, f, _ = box_mesh((-0.3, -0.3, -0.3), (0.3, 0.3, 0.3))
    ds = SDFDataset(None, 64, 10, scale_to_unit=False, mesh=(v, f))
    assert len(ds) == 10
    a, b = ds[0], ds[0]
    assert a[0].dtype == torch.float32 and a[0].shape == (64, 3) and a[1].shape == (64, 1)
    assert not torch.equal(a[0], b[0])                          # a fresh draw per item
    pts, sdf = ds.collate_fn([a, b])
    assert pts.shape == (128, 3) and sdf.shape == (128, 1) and torch.equal(pts[:64], a[0])
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.collate_fn)
    pts, sdf = next(iter(loader))
    assert pts.shape == (256, 3) and sdf.shape == (256, 1)


def test_trained_stand_in_geometries_load_and_are_the_scenes_they_were_trained_on():
    """nefii_amd/assets/scene_{bowl,frame}_sdf512.npz, scene_bowl_sdf256.npz: the conf's own SDF network trained at full width
    by the Step-1 runner on an analytic scene (tools/train_scene_sdf.py, round 5; weight_v stored in halves).  They load into
    the confs' shapes through synthetic.make_state_dict(scene='..._trained'), every effective weight keeps a full fp32 mantissa
    (the row scale g / |v| is an fp32 number: no all-zero lo fragments that would flatter the evaluators), and the
    oracle's SDF on them is the analytic scene's to the fit's accuracy, with the signs of the landmark points right."""
    import os
    import sys
    import torch
    from nefii_amd import synthetic as syn
    from oracle import nets
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import scenes
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4000, 3, generator=g) * 0.45
    x = x[x.norm(dim=1) < 1.0]
    for scene, model in (('bowl_trained', 'conf'), ('frame_trained', 'conf'), ('bowl_trained', 'neus'), ('bowl_trained', 'physg')):
        mc = syn.model_conf(model)
        sd = syn.make_state_dict(mc, seed=0, scene=scene)
        cfg = mc['implicit_network']
        with torch.no_grad():
            y = nets.sdf_forward(sd, cfg, x)[:, 0]
        t = scenes.SCENES[scene.split('_')[0]](x)
        err = (y - t).abs()
        assert err.mean().item() < 5e-4 and err.max().item() < 2e-2, (scene, model, err.mean().item(), err.max().item())
        assert ((y > 0) == (t > 0))[t.abs() > 2e-3].all()
        w, _ = nets.linear_params(sd, 'implicit_network.lin3')
        assert ((w * 64).half().float() == w * 64).float().mean().item() < 0.01      # lo halves are live
        for k, v in sd.items():
            assert torch.isfinite(v).all(), k
    with pytest.raises(FileNotFoundError):
        syn.make_state_dict(syn.model_conf('physg', hidden=128), seed=0, scene='bowl_trained')      # (no 128-wide asset)

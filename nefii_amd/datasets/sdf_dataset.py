"""SDFDataset: signed-distance samples of a mesh for the Step-1 geometry fit (reference code/datasets/sdf_dataset.py:
18-103): each item is `sample_num` positions with their signed distance, drawn afresh every time.

The reference delegates to two packages that are not installable here - trimesh 3.12.0 (mesh loading) and mesh-to-sdf
0.0.14 (`get_surface_point_cloud(...).sample_sdf_near_surface`, requirements.sh:11) - so this file restates what that
call does rather than how:
  * query positions, as mesh-to-sdf draws them: 47/50 of the samples lie near the surface - uniformly (by area) sampled
    surface points displaced by isotropic Gaussian noise, half with sigma 0.0025 and half with sigma 0.00025 - and the
    rest are uniform in the unit ball;
  * the signed distance of each query: mesh-to-sdf approximates it from a scanned point cloud (nearest of 10 M scan
    points, sign from the neighbours' normals).  Here it is the exact distance to the triangle mesh, negative inside;
    inside / outside is the parity of ray crossings, which asks for a closed mesh as the scan-based sign does.
  * `scale_to_unit`: the mesh is centred on its bounding-box centre and scaled so that its farthest vertex lies on the
    unit sphere before sampling; positions and distances are mapped back afterwards (sdf_dataset.py:52-55,62-77).
All of it is torch on whatever device the dataset is given: a 16384-sample batch against a 50 k-face mesh is ~1e9
point-triangle pairs, a few tens of milliseconds on the GPU."""
import numpy as np
import torch


def load_obj(path):
    """Wavefront OBJ -> (vertices [V, 3] float64, triangles [F, 3] int64); polygons are fan-triangulated"""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith('v '):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith('f '):
                idx = []
                for t in line.split()[1:]:
                    i = int(t.split('/')[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    if not verts or not faces:
        raise ValueError('no triangles in ' + str(path))
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


# a fixed rotation applied before the parity test, so that axis-aligned meshes do not put edges exactly under the rays
_SKEW = torch.tensor([[0.8320502943, -0.5547001962, 0.0], [0.4961389384, 0.7442084075, -0.4472135955],
                      [0.2480694692, 0.3721042038, 0.8944271910]], dtype=torch.float64)
_SKEW = torch.linalg.qr(_SKEW)[0]


class MeshSDF:
    """Exact signed distance to a closed triangle mesh, and area-uniform surface samples."""

    def __init__(self, vertices, faces, device='cpu', pair_budget=1 << 24):
        v = torch.as_tensor(vertices, dtype=torch.float64, device=device)
        f = torch.as_tensor(faces, dtype=torch.long, device=device)
        self.a, self.b, self.c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        n = torch.cross(self.b - self.a, self.c - self.a, dim=1)
        keep = n.norm(dim=1) > 0                                    # zero-area faces carry no surface
        self.a, self.b, self.c, n = self.a[keep], self.b[keep], self.c[keep], n[keep]
        self.n = n
        self.area = 0.5 * n.norm(dim=1)
        self.cdf = torch.cumsum(self.area / self.area.sum(), 0)
        self.pair_budget = pair_budget
        R = _SKEW.to(device)
        self.ra, self.rb, self.rc = self.a @ R.T, self.b @ R.T, self.c @ R.T
        self.R = R

    @property
    def device(self):
        return self.a.device

    def sample_surface(self, count, generator=None):
        u = torch.rand(count, 3, dtype=torch.float64, generator=generator).to(self.device)
        tri = torch.searchsorted(self.cdf, u[:, 0].contiguous()).clamp_(max=self.cdf.shape[0] - 1)
        r1 = u[:, 1].sqrt()
        w0, w1, w2 = 1 - r1, r1 * (1 - u[:, 2]), r1 * u[:, 2]
        return w0[:, None] * self.a[tri] + w1[:, None] * self.b[tri] + w2[:, None] * self.c[tri]

    @staticmethod
    def _segment_d2(p, a, ab):
        t = (((p - a) * ab).sum(-1) / (ab * ab).sum(-1).clamp_min(1e-300)).clamp(0, 1)
        d = p - (a + t[..., None] * ab)
        return (d * d).sum(-1)

    def __call__(self, points):
        p_all = torch.as_tensor(points, dtype=torch.float64, device=self.device).reshape(-1, 3)
        out = torch.empty(p_all.shape[0], dtype=torch.float64, device=self.device)
        F = self.a.shape[0]
        step = max(1, self.pair_budget // F)
        a, b, c, n = self.a[None], self.b[None], self.c[None], self.n[None]
        nn = (self.n * self.n).sum(-1)[None]
        for s in range(0, p_all.shape[0], step):
            p = p_all[s:s + step, None, :]                          # [P, 1, 3] against [1, F, 3]
            ap, bp, cp = p - a, p - b, p - c
            inside = ((torch.cross(b - a, ap, dim=-1) * n).sum(-1) >= 0) & \
                     ((torch.cross(c - b, bp, dim=-1) * n).sum(-1) >= 0) & \
                     ((torch.cross(a - c, cp, dim=-1) * n).sum(-1) >= 0)
            d_plane = (ap * n).sum(-1) ** 2 / nn
            d_edge = torch.minimum(torch.minimum(self._segment_d2(p, a, b - a), self._segment_d2(p, b, c - b)),
                                   self._segment_d2(p, c, a - c))
            d2 = torch.where(inside, d_plane, d_edge).min(dim=1).values
            # parity of crossings of the ray p + t * z (t > 0), in the skewed frame
            q = (p_all[s:s + step] @ self.R.T)[:, None, :]
            ra, rb, rc = self.ra[None], self.rb[None], self.rc[None]

            def edge(u, v):                                         # 2-D edge function of q against u -> v
                return (v[..., 0] - u[..., 0]) * (q[..., 1] - u[..., 1]) - (v[..., 1] - u[..., 1]) * (q[..., 0] - u[..., 0])
            e0, e1, e2 = edge(ra, rb), edge(rb, rc), edge(rc, ra)
            covers = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
            area2 = e0 + e1 + e2
            z = (e1 * ra[..., 2] + e2 * rb[..., 2] + e0 * rc[..., 2]) / torch.where(area2 == 0, torch.ones_like(area2), area2)
            crossings = (covers & (area2 != 0) & (z > q[..., 2])).sum(dim=1)
            sign = torch.where(crossings % 2 == 1, -1.0, 1.0).to(torch.float64)
            out[s:s + step] = sign * d2.sqrt()
        return out


class SDFSampler(object):                                           # sdf_dataset.py:18-77
    def __init__(self, mesh_path, number_of_points=500000, scale_to_unit=True, device='cpu', mesh=None):
        self.number_of_points = number_of_points
        self.scale_to_unit = scale_to_unit
        vertices, faces = mesh if mesh is not None else load_obj(mesh_path)
        vertices = np.asarray(vertices, dtype=np.float64)
        # centre of the bounding box / farthest vertex (trimesh's bounding_box.centroid, :66-73)
        self.center = (vertices.min(0) + vertices.max(0)) / 2 if scale_to_unit else np.zeros(3)
        self.scale = float(np.linalg.norm(vertices - self.center, axis=1).max()) if scale_to_unit else 1.0
        self.mesh_sdf = MeshSDF((vertices - self.center) / self.scale, faces, device=device)

    def sample(self, generator=None):
        n = self.number_of_points
        near = int(n * 47 / 50) // 2                                # mesh-to-sdf: surface_sample_count
        m = self.mesh_sdf
        surf = m.sample_surface(near, generator)
        noise = torch.randn(2, near, 3, dtype=torch.float64, generator=generator).to(m.device)
        ball = torch.randn(n - 2 * near, 3, dtype=torch.float64, generator=generator).to(m.device)
        radius = torch.rand(n - 2 * near, 1, dtype=torch.float64, generator=generator).to(m.device) ** (1.0 / 3.0)
        ball = ball / ball.norm(dim=1, keepdim=True) * radius
        points = torch.cat([surf + 0.0025 * noise[0], surf + 0.00025 * noise[1], ball], 0)
        sdf = m(points)
        center = torch.as_tensor(self.center, dtype=torch.float64, device=m.device)
        return (points * self.scale + center).reshape(-1, 3), (sdf * self.scale).reshape(-1, 1)


class SDFDataset(torch.utils.data.Dataset):                         # sdf_dataset.py:80-103
    def __init__(self, mesh_path, sample_num, max_iter_num, scale_to_unit=True, device='cpu', mesh=None):
        self.sample_num = sample_num
        self.max_iter_num = max_iter_num
        self.sdf_sampler = SDFSampler(mesh_path, sample_num, scale_to_unit=scale_to_unit, device=device, mesh=mesh)
        self.generator = None

    def __getitem__(self, idx):
        points, sdf = self.sdf_sampler.sample(self.generator)
        return points.float(), sdf.float()

    def __len__(self):
        return self.max_iter_num

    def collate_fn(self, batch_list):
        return tuple(torch.cat(entry, 0) for entry in zip(*batch_list))

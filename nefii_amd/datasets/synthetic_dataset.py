"""SyntheticSceneDataset: the interface IDRTrainRunner uses of the reference's SceneDataset
(code/datasets/scene_dataset.py:138-279) - per-view rgb / object mask / intrinsics / pose, per-iteration patch
sampling, the contiguous per-rank split of the patch list, sub-pixel ray jitter, collate - over procedurally generated
views.  The reference's datasets (robot, thin_cube: EXR images + cam_dict json) are not available to this build and its
image stack (imageio/OpenEXR, skimage, cv2) is not installed; everything behind __getitem__ is the same contract."""
import numpy as np
import torch

from .. import synthetic as syn


class SyntheticSceneDataset(torch.utils.data.Dataset):
    def __init__(self, gamma=1.0, data_split_dir=None, train_cameras=False, subsample=1, wo_mask=False, n_views=8,
                 img_res=(64, 64), focal=None, radius=2.4, seed=0):
        assert not train_cameras, 'camera optimisation is outside the Step-2 hot path'
        H, W = img_res
        self.img_res = [H, W]
        self.total_pixels = H * W
        self.n_cameras = n_views
        self.sampling_idx = None
        self.sampling_rays = None
        self.train_cameras = False
        g = np.random.Generator(np.random.Philox(seed))
        focal = focal if focal is not None else 1111.0 * W / 800.0
        K = np.eye(4)
        K[0, 0] = K[1, 1] = focal
        K[0, 2], K[1, 2] = W / 2.0, H / 2.0
        self.intrinsics_all, self.pose_all, self.rgb_images, self.object_masks = [], [], [], []
        for i in range(n_views):
            phi = 2 * np.pi * i / n_views
            cam = (radius * np.sin(phi) * 0.8, 0.3 * radius * np.cos(2 * phi), radius * np.cos(phi) * 0.8 + 0.6)
            self.intrinsics_all.append(torch.from_numpy(K).float())
            self.pose_all.append(torch.from_numpy(syn.look_at_origin_pose(cam)).float())
            self.rgb_images.append(torch.from_numpy(g.uniform(0.0, 1.0, size=(H * W, 3)) ** (1.0 / gamma)).float())
            m = np.ones(H * W, dtype=bool) if wo_mask else (g.uniform(size=H * W) < 0.85)
            self.object_masks.append(torch.from_numpy(m))

    def __len__(self):
        return self.n_cameras

    def __getitem__(self, idx):                                     # scene_dataset.py:149-178
        uv = np.mgrid[0:self.img_res[0], 0:self.img_res[1]].astype(np.int32)
        uv = torch.from_numpy(np.flip(uv, axis=0).copy()).float()
        uv = uv.reshape(2, -1).transpose(1, 0)
        sample = {'object_mask': self.object_masks[idx], 'uv': uv, 'intrinsics': self.intrinsics_all[idx]}
        ground_truth = {'rgb': self.rgb_images[idx]}
        if self.sampling_idx is not None:
            ground_truth['rgb'] = self.rgb_images[idx][self.sampling_idx, :]
            sample['object_mask'] = self.object_masks[idx][self.sampling_idx]
            sample['uv'] = uv[self.sampling_idx, :]
        sample['uv'] = self.ray_sample(sample['uv'])
        sample['pose'] = self.pose_all[idx]
        return idx, sample, ground_truth

    def ray_sample(self, s_uv):                                     # :180-187
        if self.sampling_rays is not None:
            s_uv = s_uv[:, None, ...] + self.sampling_rays[None, ...].to(s_uv.device)
        return s_uv

    def collate_fn(self, batch_list):                               # :196-210
        batch_list = zip(*batch_list)
        all_parsed = []
        for entry in batch_list:
            if type(entry[0]) is dict:
                all_parsed.append({k: torch.stack([obj[k] for obj in entry]) for k in entry[0].keys()})
            else:
                all_parsed.append(torch.LongTensor(entry))
        return tuple(all_parsed)

    def change_sampling_rays(self, sampling_size):                  # :212-216
        self.sampling_rays = None if sampling_size == -1 else torch.rand((sampling_size, 2)) - 0.5

    def change_sampling_idx(self, sampling_size):                   # :218-222
        self.sampling_idx = None if sampling_size == -1 else torch.randperm(self.total_pixels)[:sampling_size]

    def change_sampling_idx_patch(self, N_patch, r_patch=1):        # :224-251
        if N_patch == -1:
            self.sampling_idx = None
            return
        H, W = self.img_res
        u, v = np.meshgrid(np.arange(-r_patch, r_patch), np.arange(-r_patch, r_patch))
        offsets = v.reshape(-1) * W + u.reshape(-1)
        u, v = np.meshgrid(np.arange(r_patch, W - r_patch), np.arange(r_patch, H - r_patch))
        u, v = u.reshape(-1), v.reshape(-1)
        select = np.random.choice(u.shape[0], size=(N_patch,), replace=False)
        select = v[select] * W + u[select]
        select = np.stack([select + shift for shift in offsets], axis=1).reshape(-1)
        self.sampling_idx = torch.from_numpy(select).long()

    def scatter_sampling_idx_patch(self, rank, world_size, N_patch, r_patch=1):     # :268-279
        if self.sampling_idx is None:
            return
        sel = self.sampling_idx.reshape(-1, 4 * r_patch * r_patch)
        sub = sel.shape[0] // world_size
        sel = sel[rank * sub: rank * sub + sub] if rank < world_size - 1 else sel[rank * sub:]
        self.sampling_idx = sel.reshape(-1)

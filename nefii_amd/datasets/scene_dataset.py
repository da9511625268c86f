"""SceneDataset: the data the Step-2 iteration is fed from (reference code/datasets/scene_dataset.py:11-279): an instance
directory with `cam_dict_norm.json` (per image: K and W2C as flattened 4x4), `image/*.{exr,png,jpg}` (linear radiance
in EXR, 8-bit otherwise; raised to `gamma`) and `mask/*.png` (object mask = luma > 0.5), all views held in host
memory as [H*W, 3] / [H*W] tensors; per-iteration patch / pixel index sampling, the contiguous per-rank split of the
index list, sub-pixel ray jitter and the collate function of the training DataLoader.

Same constructor, attributes and methods as the reference class, so `IDRTrainRunner` and the reference's own runner
take either.  Differences, all in the image stack the reference borrows (imageio/freeimage, torchvision - neither is
installable here): EXR files go through utils/exr.py, 8-bit files through Pillow, and `subsample` resizes with
`torch.nn.functional.interpolate(mode='bilinear', antialias=True)`, the op torchvision's `Resize(antialias=True)` calls
on float tensors."""
import json
import os

import numpy as np
import torch

from ..utils import general as utils
from ..utils import rend_util


def read_cam_dict(cam_dict_file):                                   # scene_dataset.py:11-22
    with open(cam_dict_file) as fp:
        cam_dict = json.load(fp)
    for x in sorted(cam_dict.keys()):
        K = np.array(cam_dict[x]['K']).reshape((4, 4))
        W2C = np.array(cam_dict[x]['W2C']).reshape((4, 4))
        cam_dict[x]['K'] = K
        cam_dict[x]['W2C'] = W2C
        cam_dict[x]['C2W'] = np.linalg.inv(W2C)
    return cam_dict


class SceneDataset(torch.utils.data.Dataset):
    def __init__(self, gamma, instance_dir, train_cameras, subsample=1, wo_mask=False):     # :28-103
        self.instance_dir = instance_dir
        assert os.path.exists(self.instance_dir), 'Data directory is empty'
        self.gamma = gamma
        self.train_cameras = train_cameras
        self.subsample = subsample

        image_paths = sorted(utils.glob_imgs(os.path.join(self.instance_dir, 'image')))
        mask_paths = sorted(utils.glob_imgs(os.path.join(self.instance_dir, 'mask')))
        cam_dict = read_cam_dict(os.path.join(self.instance_dir, 'cam_dict_norm.json'))
        print('Found # images, # masks, # cameras: ', len(image_paths), len(mask_paths), len(cam_dict))
        self.n_cameras = len(image_paths)
        self.image_paths = image_paths
        self.single_imgname = None
        self.single_imgname_idx = None
        self.sampling_idx = None
        self.sampling_rays = None

        self.intrinsics_all, self.pose_all = [], []
        for x in sorted(cam_dict.keys()):
            self.intrinsics_all.append(torch.from_numpy(cam_dict[x]['K'].astype(np.float32)).float())
            self.pose_all.append(torch.from_numpy(cam_dict[x]['C2W'].astype(np.float32)).float())

        if len(image_paths) > 0:
            self.has_groundtruth = True
            self.rgb_images = []
            print('Applying inverse gamma correction: ', self.gamma)
            for path in image_paths:
                rgb = np.power(rend_util.load_rgb(path), self.gamma)
                H, W = rgb.shape[1:3]
                self.img_res = [H, W]
                self.total_pixels = H * W
                self.rgb_images.append(torch.from_numpy(rgb.reshape(3, -1).transpose(1, 0).copy()).float())
        else:
            # no ground truth: the resolution comes from the first camera's normalised intrinsics (:81-89; the reference
            # indexes `cam_dict.values()[0]`, a Python-2 idiom that raises on Python 3 - the intent is kept)
            self.has_groundtruth = False
            self.n_cameras = len(cam_dict)
            K = cam_dict[sorted(cam_dict.keys())[0]]['K']
            W, H = int(2. / K[0, 0]), int(2. / K[1, 1])
            print('No ground-truth images available. Image resolution of predicted images: ', H, W)
            self.img_res = [H, W]
            self.total_pixels = H * W
            self.rgb_images = [torch.ones((self.total_pixels, 3), dtype=torch.float32)] * self.n_cameras

        if len(mask_paths) > 0 and not wo_mask:
            assert len(mask_paths) == self.n_cameras
            self.object_masks = [torch.from_numpy(rend_util.load_mask(path).reshape(-1)).bool() for path in mask_paths]
        else:
            self.object_masks = [torch.ones((self.total_pixels,)).bool()] * self.n_cameras

        if self.subsample is not None and self.subsample != 1:
            print('resizing data with subsample=', self.subsample)
            self.resize()

    def resize(self):                                               # :105-136
        old = (self.img_res[0], self.img_res[1])
        new = (int(old[0] * self.subsample), int(old[1] * self.subsample))
        self.img_res = [new[0], new[1]]
        self.total_pixels = new[0] * new[1]
        scale = max(new) / max(old)
        for K in self.intrinsics_all:
            K[0, 0] *= scale
            K[0, 2] *= scale
            K[1, 1] *= scale
            K[1, 2] *= scale

        def resizer(img):                                           # [C, H, W] float
            return torch.nn.functional.interpolate(img[None], size=new, mode='bilinear', antialias=True,
                                                   align_corners=False)[0]

        shared = {}                                                 # views may share one placeholder tensor
        for i in range(len(self.rgb_images)):
            key = id(self.rgb_images[i])
            if key not in shared:
                img = self.rgb_images[i].reshape(old[0], old[1], 3).permute(2, 0, 1)
                shared[key] = resizer(img).reshape(3, -1).transpose(1, 0).float()
            self.rgb_images[i] = shared[key]
        shared = {}
        for i in range(len(self.object_masks)):
            key = id(self.object_masks[i])
            if key not in shared:
                m = resizer(self.object_masks[i].float().reshape(1, old[0], old[1])) > 0.5
                shared[key] = m.reshape(-1).bool()
            self.object_masks[i] = shared[key]

    def __len__(self):
        return self.n_cameras

    def return_single_img(self, img_name):                          # :141-147
        self.single_imgname = img_name
        for idx in range(len(self.image_paths)):
            if os.path.basename(self.image_paths[idx]) == self.single_imgname:
                self.single_imgname_idx = idx
                break
        print('Always return: ', self.single_imgname, self.single_imgname_idx)

    def __getitem__(self, idx):                                     # :149-178
        if self.single_imgname_idx is not None:
            idx = self.single_imgname_idx
        uv = np.mgrid[0:self.img_res[0], 0:self.img_res[1]].astype(np.int32)
        uv = torch.from_numpy(np.flip(uv, axis=0).copy()).float()
        uv = uv.reshape(2, -1).transpose(1, 0)                      # [H*W, 2] as (x, y)
        sample = {'object_mask': self.object_masks[idx], 'uv': uv, 'intrinsics': self.intrinsics_all[idx]}
        ground_truth = {'rgb': self.rgb_images[idx]}
        if self.sampling_idx is not None:
            ground_truth['rgb'] = self.rgb_images[idx][self.sampling_idx, :]
            sample['object_mask'] = self.object_masks[idx][self.sampling_idx]
            sample['uv'] = uv[self.sampling_idx, :]
        sample['uv'] = self.ray_sample(sample['uv'])
        if not self.train_cameras:
            sample['pose'] = self.pose_all[idx]
        return idx, sample, ground_truth

    def ray_sample(self, s_uv):                                     # :180-187
        if self.sampling_rays is not None:
            s_uv = s_uv[:, None, ...] + self.sampling_rays[None, ...].to(s_uv.device)      # [S, R, 2]
        return s_uv

    def batch_ray_sample(self, s_uv_batch):                         # :189-194
        B, S, _ = s_uv_batch.shape
        return self.ray_sample(s_uv_batch.reshape(B * S, 2)).reshape(B, S, -1, 2)

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
        """N_patch patches of (2 r_patch) x (2 r_patch) pixels, centres drawn without replacement"""
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

    def get_pose_init(self):                                        # :253-258 (camera optimisation: out of scope)
        raise NotImplementedError('camera optimisation (train_cameras) is outside the Step-2 hot path')

    def scatter_sampling_idx(self, rank, world_size):               # :260-266
        if self.sampling_idx is not None:
            sub = self.sampling_idx.shape[0] // world_size
            self.sampling_idx = (self.sampling_idx[rank * sub: rank * sub + sub] if rank < world_size - 1
                                 else self.sampling_idx[rank * sub:])

    def scatter_sampling_idx_patch(self, rank, world_size, N_patch, r_patch=1):     # :268-279
        if self.sampling_idx is not None:
            sel = self.sampling_idx.reshape(-1, 4 * r_patch * r_patch)
            sub = sel.shape[0] // world_size
            sel = sel[rank * sub: rank * sub + sub] if rank < world_size - 1 else sel[rank * sub:]
            self.sampling_idx = sel.reshape(-1)

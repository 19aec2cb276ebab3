"""dataloaders.py -- sample source for the trainer.

The reference's KITTI loader (dataloaders.py:14-252: PIL decode, torchvision transforms, calib/oxts parsing) is host
I/O outside this round's hot path ("next" row 1) and KITTI is not available offline.  What the
training step consumes is kept: a dict with 'tgt' [3,H,W], 'ref_imgs' [2 x [3,H,W]], 'intrinsics' [3,3] fp64,
'groundtruth' [1,H,W] (reference dataloaders.py:226-251).  SyntheticTriplets produces such samples from a seed.
"""
import torch
from torch.utils.data import Dataset


class SyntheticTriplets(Dataset):
    def __init__(self, config, transforms=None, length=None):
        aug = config['datasets']['augmentation']
        self.H, self.W = aug['image_height'], aug['image_width']
        self.length = length or int(config['datasets'].get('synthetic_length', 64))
        self.seed = int(config['action'].get('random_seed', 0))

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        H, W = self.H, self.W
        imgs = []
        for _ in range(3):
            x = torch.randn(1, 3, H, W, generator=g)
            x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)[0]
            imgs.append(x.contiguous())
        K = torch.tensor([[0.58 * W, 0.0, 0.5 * W], [0.0, 1.92 * H, 0.5 * H], [0.0, 0.0, 1.0]], dtype=torch.float64)
        return {"tgt": imgs[0], "ref_imgs": [imgs[1], imgs[2]], "intrinsics": K, "groundtruth": torch.zeros(1, H, W)}


def UnSupKittiDataset(config, transforms=None):
    if config['datasets'].get('dataset', ['KITTI']) == ['synthetic']:
        return SyntheticTriplets(config, transforms)
    raise NotImplementedError("the KITTI file loader is outside this round's hot path (SURVEY.md 8f); set "
                              "datasets.dataset: ['synthetic'] or pass dataset= to Trainer")

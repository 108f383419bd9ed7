"""Seeded synthetic ScanNet-shaped inputs shared by the tests."""
import torch


def scene(b, n, seed=0, dup=0, zero_tail=0):
    """xyz ~ U([0,8]x[0,8]x[0,3]) m; optional duplicated points and an all-zero padded tail
    (exercise FPS tie-break / skip rules, sampling_gpu.cu:100-110)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(b, n, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    if dup:
        src = torch.randint(0, n, (b, dup), generator=g)
        dst = torch.randint(0, n, (b, dup), generator=g)
        for i in range(b):
            xyz[i, dst[i]] = xyz[i, src[i]]
    if zero_tail:
        xyz[:, n - zero_tail:] = 0.0
    return xyz.contiguous()


def feats(b, c, n, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, c, n, generator=g).contiguous()


def poses(b, seed=2):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(b, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    ang = (torch.rand(b, generator=g) * 2 - 1) * 3.14159265
    q = torch.stack([torch.zeros(b), torch.zeros(b), torch.sin(ang / 2), torch.cos(ang / 2)], 1)
    return torch.cat([t, q], 1).contiguous()

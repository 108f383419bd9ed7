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


def tiny_head(din=12, hidden=16, num_answers=9):
    """The model of tests/golden/harness_trajectory.npz (make_golden_r2.py::TinyHead): parameter names
    cover both optimizer groups ("bias" / "LayerNorm.weight" vs the rest, train.py:186)."""
    import torch.nn as nn
    import torch.nn.functional as F

    class TinyHead(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(din, hidden)
            self.LayerNorm = nn.LayerNorm(hidden)
            self.aux_reg = nn.Linear(hidden, 7)
            self.answer_cls = nn.Linear(hidden, num_answers)

        def forward(self, data_dict):
            h = F.gelu(self.LayerNorm(self.proj(data_dict["x"])))
            data_dict["aux_scores"] = self.aux_reg(h)
            data_dict["answer_scores"] = self.answer_cls(h)
            return data_dict

    return TinyHead()

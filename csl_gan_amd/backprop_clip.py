"""Per-sample L2 clip primitive of the reference's experimental backprop clipping
(backprop_clip.py:18-22).  The PGCWrapper / BackpropClipper bookkeeping (backprop_clip.py:49-158) is
not carried: it is self-declared unfinished (options.py:243-244), hard-codes a 1x28x28 input
(backprop_clip.py:123) and needs torchinfo; SURVEY.md §2 row 10 marks it low priority."""
import torch

from . import ops


def l2_clip(t, C):
    """Rows (samples) whose L2 norm over all non-batch dims exceeds C are rescaled to norm C."""
    if t.is_cuda:
        if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
            v = t.permute(0, 2, 3, 1)
            return ops.l2_clip_rows(v, C).permute(0, 3, 1, 2)
        return ops.l2_clip_rows(t.contiguous(), C)
    dims = tuple(range(1, t.dim()))
    norm = t.norm(2, dim=dims, keepdim=True)
    return torch.where(norm > C, C * (t / norm), t)

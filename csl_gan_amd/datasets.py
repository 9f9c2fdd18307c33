"""Real-data loaders with the reference's contract (datasets.py:20-63, init_util.py:13-42), written
without torchvision (absent here): PIL + numpy do the decode / resize / crop / flip / normalise, and raw
MNIST idx files are parsed directly.  SURVEY.md §8f item 4 — the benchmark and the tests use synthetic
tensors; these classes exist so the CLI can train on the real datasets when their files are present.

CelebADataset: images "<root>/NNNNNN.jpg" numbered from 1 (+offset); transform = resize shorter side to
im_size (bilinear), centre crop im_size x im_size, random horizontal flip (p=0.5), scale to [0,1], normalise
with mean 0.5 / std 0.5; binary label from one column of list_attr_celeba.txt (value 1 -> class 1).
"""
import gzip
import os
import struct

import numpy as np
import torch
from torch.utils.data import Dataset

try:
    from PIL import Image
except Exception:          # pragma: no cover - PIL is present in the image
    Image = None


def _load_celeba_attr(attr_file, attr, first, count):
    """Column `attr` of rows [first, first+count) of list_attr_celeba.txt as 0/1 ints (header: 2 lines)."""
    with open(attr_file) as f:
        f.readline()
        names = f.readline().split()
        col = names.index(attr) + 1          # +1: the file-name column
        vals = []
        for i, line in enumerate(f):
            if i < first:
                continue
            if len(vals) >= count:
                break
            vals.append(1 if int(line.split()[col]) == 1 else 0)
    return np.asarray(vals, dtype=np.int64)


class CelebADataset(Dataset):
    def __init__(self, root, im_size=32, length=None, offset=0, ext="jpg", attr_file=None, attr=None, flip=True, seed=None):
        if Image is None:
            raise RuntimeError("PIL is required to read CelebA images")
        self.root, self.im_size, self.offset, self.ext, self.flip = root, im_size, offset, ext, flip
        self.length = length if length else len(os.listdir(root))
        self.rng = np.random.default_rng(seed)
        if attr is None:
            self.labels, self.label_true_count = None, None
        else:
            self.labels = _load_celeba_attr(attr_file, attr, offset, self.length)
            self.label_true_count = int((self.labels == 1).sum())
        self.n_classes = 1

    def __len__(self):
        return self.length

    def _transform(self, img):
        s = self.im_size
        w, h = img.size
        if w <= h:
            nw, nh = s, max(s, int(round(h * s / w)))
        else:
            nw, nh = max(s, int(round(w * s / h))), s
        img = img.resize((nw, nh), Image.BILINEAR)
        left, top = (nw - s) // 2, (nh - s) // 2
        a = np.asarray(img.crop((left, top, left + s, top + s)), dtype=np.float32) / 255.0
        if self.flip and self.rng.random() < 0.5:
            a = a[:, ::-1]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return (t - 0.5) / 0.5

    def __getitem__(self, index):
        number = index + 1
        path = os.path.join(self.root, str(self.offset + number).zfill(6) + "." + self.ext)
        img = self._transform(Image.open(path).convert("RGB"))
        label = 0 if self.labels is None else int(self.labels[number - 1])
        return img, label

    def get_item_with_label(self, label, number=None):
        number = int(self.rng.integers(0, self.length)) if number is None else number
        while self.labels[number] != label:
            number = (number + 1) % self.length
        return self.__getitem__(number)


def _read_idx(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        magic, = struct.unpack(">I", f.read(4))
        ndim = magic & 0xFF
        dims = struct.unpack(">" + "I" * ndim, f.read(4 * ndim))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(dims)


class MNISTDataset(Dataset):
    """Raw idx files (train-images-idx3-ubyte[.gz] ...) under <root>/MNIST/raw or <root>; ToTensor scaling to
    [0,1]; `per_class` keeps the first train_set_size/10 images of every digit (init_util.py:19-23)."""

    def __init__(self, root, train=True, per_class=None):
        stem = "train" if train else "t10k"
        base = None
        for d in (os.path.join(root, "MNIST", "raw"), root):
            for suf in ("", ".gz"):
                if os.path.exists(os.path.join(d, stem + "-images-idx3-ubyte" + suf)):
                    base = (d, suf)
        if base is None:
            raise FileNotFoundError("MNIST idx files not found under %s" % root)
        d, suf = base
        x = _read_idx(os.path.join(d, stem + "-images-idx3-ubyte" + suf)).astype(np.float32) / 255.0
        y = _read_idx(os.path.join(d, stem + "-labels-idx1-ubyte" + suf)).astype(np.int64)
        if per_class is not None:
            keep = np.concatenate([np.nonzero(y == c)[0][:per_class] for c in range(10)])
            x, y = x[keep], y[keep]
        self.x, self.y = torch.from_numpy(x).unsqueeze(1), torch.from_numpy(y)

    def __len__(self):
        return len(self.y)

    def __getitem__(self, i):
        return self.x[i], int(self.y[i])

    def get_item_with_label(self, label, number=None):
        idx = torch.nonzero(self.y == int(label)).flatten()
        i = int(idx[torch.randint(0, len(idx), (1,))])
        return self.__getitem__(i)

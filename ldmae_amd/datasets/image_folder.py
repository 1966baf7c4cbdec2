"""``root/<class>/<image>`` reader with the sample order and labels of torchvision's ImageFolder (classes = sorted directory names,
samples = sorted walk per class, the same extension list), on PIL alone -- torchvision is not a dependency of this package.
Used by extract_features.py (reference: extract_features.py:103-108).  Host I/O only."""
import os

from torch.utils.data import Dataset

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class ImageFolder(Dataset):
    def __init__(self, root, transform=None):
        self.root, self.transform = root, transform
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for d, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(d, f), self.class_to_idx[c]))
        self.targets = [t for _, t in self.samples]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        from PIL import Image
        path, target = self.samples[i]
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB")
        return (self.transform(img) if self.transform is not None else img), target

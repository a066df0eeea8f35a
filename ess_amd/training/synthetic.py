"""Seeded synthetic batches in the shapes the ESS loaders produce: events = randn * (rand < 0.1)
voxel grids [B, T*C, H, W], images rand [B, 1, H, W], labels randint(0, K) with an ignore (255) band.  Stands in
for datasets/* (CPU voxel-grid generation, out of scope) wherever `synthetic.enabled` is set."""
import torch


def make_batch(B, T, C, H, W, K, seed, device, sparsity=0.1):
    g = torch.Generator().manual_seed(int(seed))
    ev = torch.randn(B, T * C, H, W, generator=g) * (torch.rand(B, T * C, H, W, generator=g) < sparsity).float()
    img = torch.rand(B, 1, H, W, generator=g)
    lab_a = torch.randint(0, K, (B, H, W), generator=g)
    lab_b = torch.randint(0, K, (B, H, W), generator=g)
    lab_a[:, : max(1, H // 16)] = 255
    lab_b[:, -max(1, H // 16):] = 255
    return [t.to(device) for t in (ev, img, lab_a, lab_b)]


class SyntheticPairedLoader:
    """Iterates `steps` batches of [[img, label_a], [events, label_b]] (the WrapperDataset layout,
    datasets/wrapper_dataloader.py:53-54) resident on the device; rank-dependent seeds for data parallelism.
    events_only / images_only: the single-sensor [data, label] layout of the validation loaders."""

    def __init__(self, steps, B_a, B_b, T, C, H, W, K, device, seed=0, events_only=False, images_only=False):
        self.steps, self.args, self.device, self.seed = steps, (T, C, H, W, K), device, seed
        self.B_a, self.B_b, self.events_only, self.images_only = B_a, B_b, events_only, images_only

    def __len__(self):
        return self.steps

    def createIterators(self):
        pass

    def __iter__(self):
        T, C, H, W, K = self.args
        for i in range(self.steps):
            ev, img, lab_a, lab_b = make_batch(max(self.B_a, self.B_b), T, C, H, W, K, self.seed + i, self.device)
            if self.events_only:
                yield [ev[:self.B_b], lab_b[:self.B_b]]
            elif self.images_only:
                yield [img[:self.B_a], lab_a[:self.B_a]]
            else:
                yield [[img[:self.B_a], lab_a[:self.B_a]], [ev[:self.B_b], lab_b[:self.B_b]]]

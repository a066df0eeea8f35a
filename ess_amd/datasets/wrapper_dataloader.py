"""Two-loader zipper of the UDA trainer (reference: datasets/wrapper_dataloader.py).

Same constructor, `__len__`, `createIterators()` and `__getitem__` contract: one item = ([a tensors...], [b tensors...]) moved to
`device`; the LONGER loader (or the one `dataset_len_to_use` names) sets the epoch length and is consumed once, the shorter one
restarts whenever it runs out.  Whether a side yields (data, label) or (data, paired_data, label) follows its dataset's
`require_paired_data`; the tuples are forwarded as they come, so the four paired / unpaired combinations need no separate code."""
from torch.utils.data import Dataset


class WrapperDataset(Dataset):
    def __init__(self, dataloader_a, dataloader_b, device, dataset_len_to_use=None):
        self.dataloader_a, self.dataloader_b = dataloader_a, dataloader_b
        self.require_paired_data_a = dataloader_a.dataset.require_paired_data
        self.require_paired_data_b = dataloader_b.dataset.require_paired_data
        self.device = device
        self.dataset_a_larger = len(dataloader_a) > len(dataloader_b)
        if dataset_len_to_use == 'first':
            self.dataset_a_larger = True
        elif dataset_len_to_use == 'second':
            self.dataset_a_larger = False
        self.createIterators()

    def __len__(self):
        return len(self.dataloader_a if self.dataset_a_larger else self.dataloader_b)

    def createIterators(self):
        self.dataloader_a_iter = iter(self.dataloader_a)
        self.dataloader_b_iter = iter(self.dataloader_b)

    def _next_restarting(self, side):
        try:
            return next(getattr(self, f'dataloader_{side}_iter'))
        except StopIteration:
            setattr(self, f'dataloader_{side}_iter', iter(getattr(self, f'dataloader_{side}')))
            return next(getattr(self, f'dataloader_{side}_iter'))

    def __getitem__(self, idx):
        """-> ([a...], [b...]); the index is ignored (the loaders' own order is what counts), as in the reference.  The shorter
        side is advanced first, the epoch-setting side raises StopIteration at the end of the epoch."""
        if self.dataset_a_larger:
            b = self._next_restarting('b')
            a = next(self.dataloader_a_iter)
        else:
            a = self._next_restarting('a')
            b = next(self.dataloader_b_iter)
        want_a, want_b = 3 if self.require_paired_data_a else 2, 3 if self.require_paired_data_b else 2
        if len(a) != want_a or len(b) != want_b:
            raise ValueError(f'WrapperDataset: loader a / b yielded {len(a)} / {len(b)} tensors, expected {want_a} / {want_b}')
        return [t.to(self.device) for t in a], [t.to(self.device) for t in b]

"""Stub loaders + the drive loop shared by tests/golden/make_golden_zipper.py (which runs the REFERENCE class here in the build
container) and tests/test_host_cpu.py (which runs ess_amd's class against the committed fixture)."""
import torch


class StubLoader:
    def __init__(self, n, paired, tag):
        self.n, self.tag = n, tag
        self.dataset = type('D', (), {'require_paired_data': paired})()

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            t = torch.tensor([self.tag * 100 + i])
            yield (t, t + 1000, t + 2000) if self.dataset.require_paired_data else (t, t + 2000)


def run(cls, na, nb, pa, pb, use, epochs=2):
    w = cls(StubLoader(na, pa, 1), StubLoader(nb, pb, 2), 'cpu', use)
    log = {'len': len(w), 'epochs': []}
    for _ in range(epochs):
        w.createIterators()
        items = []
        for i in range(len(w) + 1):  # one past the end: the epoch-setting loader must raise StopIteration
            try:
                a, b = w[i]
                items.append([[int(t) for t in a], [int(t) for t in b]])
            except StopIteration:
                items.append('stop')
                break
        log['epochs'].append(items)
    return log

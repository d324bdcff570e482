class AverageMeter:
    """Running average with `.update(val, n)` / `.avg` (what trainers/base.py:280 expects from utils.misc)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.sum = self.avg = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = float(val)
        self.sum += float(val) * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)

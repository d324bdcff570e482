"""`utils.misc`: the running-average meters the trainers fill (reference trainers/base.py:275-307 expects `AverageMeter` with
`.update(val, n)` / `.avg` from the un-vendored `utils.misc`).

`AverageMeter` is the reference's host-side meter.  `DeviceMeters` keeps the same sums ON THE GPU while a batch loop runs: the
reference reads every loss term and accuracy back with `.item()` once per batch (>= 7 host synchronisations per batch,
trainers/coop.py:80-101), which drains the launch queue of a 12 ms encode step each time; here a batch enqueues two tiny launches
(stack + add) and the host reads the sums ONCE per epoch (or as asynchronous snapshots for progress lines)."""
from __future__ import annotations

from collections import OrderedDict


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


class _Slot:
    """`meters[name]` of a DeviceMeters: `.update(val, n)` like AverageMeter; a tensor `val` never leaves the device."""

    def __init__(self, owner, name):
        self._owner, self._name = owner, name

    def update(self, val, n=1):
        self._owner.update_many({self._name: val}, n)

    @property
    def avg(self):                       # synchronises: for callers outside the batch loop
        return self._owner.finalize()[self._name].avg


class DeviceMeters:
    """name -> sample-weighted running sum, accumulated in float64 on `device`; counts (batch sizes) are host integers.

    update_many({name: 0-dim tensor or float}, n): sum[name] += value * n, no host synchronisation.
    finalize() -> OrderedDict name -> AverageMeter (ONE device -> host copy).
    snapshot() / latest(): asynchronous copy of the sums into pinned memory for progress lines -- `latest()` returns the averages of
    the newest snapshot whose copy has completed (never waits).
    `stream` (optional, a torch.cuda.Stream): every device operation of the meters runs on it -- the evaluation loop computes the loss
    terms and accuracies of batch i there, beside the encode of batch i + 1 (trainers/coop.py: inference_one_batch)."""

    CAP = 64

    def __init__(self, device):
        import torch
        self._torch = torch
        self.device = torch.device(device)
        self._idx = OrderedDict()        # name -> slot
        self._count = []                 # per slot, host
        self._sums = torch.zeros(self.CAP, dtype=torch.float64, device=self.device)
        self._host = {}                  # float values given by the caller: plain host sums
        self._snaps = []                 # (event, pinned tensor, counts copy, names copy)
        self.stream = None

    def _on_stream(self):
        import contextlib
        return self._torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def __getitem__(self, name):
        return _Slot(self, name)

    def __contains__(self, name):
        return name in self._idx or name in self._host

    def _slot(self, name):
        if name not in self._idx:
            if len(self._idx) >= self.CAP:
                raise RuntimeError(f"more than {self.CAP} device meters")
            self._idx[name] = len(self._idx)
            self._count.append(0)
        return self._idx[name]

    def update_many(self, values, n=1):
        torch = self._torch
        dev_names, dev_vals = [], []
        for name, v in values.items():
            if torch.is_tensor(v):
                dev_names.append(name)
                dev_vals.append(v.detach().reshape(()).to(torch.float32))
            else:                        # already a host number: no reason to ship it to the device
                s, c = self._host.get(name, (0.0, 0))
                self._host[name] = (s + float(v) * n, c + n)
        if not dev_vals:
            return
        slots = [self._slot(nm) for nm in dev_names]
        with self._on_stream():
            vals = torch.stack(dev_vals)                                       # one launch
            if slots == list(range(slots[0], slots[0] + len(slots))):          # the usual case: same meters, same order, every batch
                self._sums[slots[0]:slots[0] + len(slots)].add_(vals, alpha=float(n))   # one launch (fp32 -> fp64 promotion in place)
            else:
                self._sums.index_add_(0, torch.tensor(slots, device=self.device), vals.double() * float(n))
        for s in slots:
            self._count[s] += n

    # ---- reading ---------------------------------------------------------------------------------------------------------------
    def _to_meters(self, sums, counts, names, host):
        out = OrderedDict()
        for name, s in names.items():
            m = AverageMeter()
            m.sum, m.count = float(sums[s]), int(counts[s])
            m.avg = m.sum / max(m.count, 1)
            m.val = m.avg
            out[name] = m
        for name, (s, c) in host.items():
            m = out.get(name) or AverageMeter()
            m.sum, m.count = m.sum + s, m.count + c
            m.avg = m.sum / max(m.count, 1)
            out[name] = m
        return out

    def finalize(self):
        with self._on_stream():
            sums = self._sums[:max(1, len(self._idx))].cpu() if self._idx else []
        return self._to_meters(sums, self._count, self._idx, self._host)

    def snapshot(self):
        torch = self._torch
        if not self._idx or self.device.type != "cuda":
            return
        k = len(self._idx)
        pinned = torch.empty(k, dtype=torch.float64, pin_memory=True)
        with self._on_stream():
            pinned.copy_(self._sums[:k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        self._snaps.append((ev, pinned, list(self._count), OrderedDict(self._idx), dict(self._host)))
        self._snaps = self._snaps[-4:]

    def latest(self):
        for ev, pinned, counts, names, host in reversed(self._snaps):
            if ev.query():
                return {k: round(m.avg, 4) for k, m in self._to_meters(pinned, counts, names, host).items()}
        return None


def module_to_device(module, device):
    """`module.to(device)` for a host module whose tensors are many: ONE host -> device copy per dtype instead of one per tensor.

    On ROCm every copy from fresh pageable memory registers its pages with the driver first -- ~1.3 ms per call whatever the size: the
    567 tensors of a ViT-B/16 ConceptHash model took 0.78 s to move, a single 344 MB tensor takes 17 ms.  The tensors are packed into
    one host buffer per dtype (each at a 256-byte boundary), copied once, and every parameter / buffer becomes a view of the device
    buffer (`param.data = view`, exactly what `nn.Module._apply` does with the tensor `.to` returns); `module.to(device)` then runs as
    usual -- nothing left to copy -- so hooks in `_apply` overrides still fire.  Other moves (device -> device, dtype changes) are
    `module.to` unchanged."""
    import torch
    device = torch.device(device)
    if device.type != "cuda":
        return module.to(device)
    seen, groups = set(), {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t is None or id(t) in seen or t.device.type != "cpu" or t.numel() == 0 or not t.data.is_contiguous():
            continue
        seen.add(id(t))
        groups.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype, tensors in groups.items():
            if len(tensors) < 2:
                continue
            align = max(1, 256 // tensors[0].element_size())
            offsets, total = [], 0
            for t in tensors:
                offsets.append(total)
                total += -(-t.numel() // align) * align
            host = torch.empty(total, dtype=dtype)
            for t, o in zip(tensors, offsets):
                host[o:o + t.numel()].copy_(t.data.reshape(-1))
            dev = host.to(device)
            for t, o in zip(tensors, offsets):
                t.data = dev[o:o + t.numel()].view(t.shape)
    return module.to(device)

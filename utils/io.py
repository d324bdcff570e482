"""`utils.io` (un-vendored in the reference; call sites experiments/test_hashing.py:21,174,180): an asynchronous
save queue so that writing `outputs.pth` does not stall the evaluation."""
from __future__ import annotations

import os
import queue
import threading

import torch

_q = None
_worker = None


def _run():
    while True:
        item = _q.get()
        if item is None:
            _q.task_done()
            return
        obj, path = item
        try:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp"     # unique per process: two ranks never share a temporary name
            torch.save(obj, tmp)
            os.replace(tmp, path)
        finally:
            _q.task_done()


def init_save_queue():
    global _q, _worker
    if _q is None:
        _q = queue.Queue()
        _worker = threading.Thread(target=_run, daemon=True)
        _worker.start()


def fast_save(obj, path):
    if _q is None:
        init_save_queue()
    _q.put((obj, path))


def join_save_queue():
    global _q, _worker
    if _q is not None:
        _q.put(None)
        _q.join()
        _worker.join()
        _q = _worker = None

"""Accuracy helpers (reference utils/metrics.py:1-29): top-1 / top-k accuracy on logits and on a Hamming-distance
matrix versus a class codebook.  Tiny (B, C) bookkeeping for the meters."""
import torch


def calculate_accuracy(logits, labels, onehot=True, multiclass=False, topk=1):
    if multiclass or topk != 1:
        k = 5 if topk == 1 else topk
        pred = logits.topk(k, 1, True, True)[1]
        target = labels.argmax(1) if onehot else labels
        return pred.eq(target.reshape(-1, 1)).any(1).float().sum(0, keepdim=True) / logits.size(0)
    if labels.dim() == 2:
        labels = labels.argmax(1)
    return (logits.argmax(1) == labels).float().mean()


def calculate_accuracy_hamm_dist(hamm_dist, labels, onehot=True, multiclass=False):
    if multiclass:
        pred = hamm_dist.topk(5, 1, False, True)[1]
        return pred.eq(labels.argmax(1).reshape(-1, 1)).any(1).float().sum(0, keepdim=True) / hamm_dist.size(0)
    if labels.dim() == 2:
        labels = labels.argmax(1)
    return (hamm_dist.argmin(1) == labels).float().mean()

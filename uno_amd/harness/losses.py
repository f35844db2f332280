import torch


def lp_loss_rel_sum(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Sum over the batch of ||pred_b - target_b||_2 / ||target_b||_2 - what the reference trains on:
    LpLoss(size_average=False)(out, y) (utilities3.py:86-100, train_darcy.py:42,53)."""
    n = pred.shape[0]
    diff = torch.linalg.vector_norm(pred.reshape(n, -1) - target.reshape(n, -1), ord=2, dim=1)
    return (diff / torch.linalg.vector_norm(target.reshape(n, -1), ord=2, dim=1)).sum()

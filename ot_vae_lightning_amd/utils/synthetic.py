"""Synthetic inputs for benchmarks / smoke runs (no dataset or network on the GPU box): SURVEY.md section 8(d) C1."""
import torch


def mnist_like(batch: int, seed: int = 42) -> torch.Tensor:
    """[B,1,32,32] fp32: ~19% 'ink' pixels U[0,1), MNIST-normalised, then zero-padded 28 -> 32 (border = 0.0) the way
    the reference's MNIST32 datamodule does (data/__init__.py:37)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(batch, 1, 28, 28, generator=g)
    m = torch.rand(batch, 1, 28, 28, generator=g)
    x = torch.where(m < 0.81, torch.zeros_like(u), u)
    x = (x - 0.1307) / 0.3081
    return torch.nn.functional.pad(x, (2, 2, 2, 2))

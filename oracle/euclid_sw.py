"""oracle/euclid_sw.py -- CPU restatement (torch, float64-capable) of the notebooks' Euclidean sliced-Wasserstein
(/root/reference/Wasserstein_flow_problem/Flow_cube.ipynb:280-292).  TEST INFRASTRUCTURE ONLY.
The notebook cell cannot be imported (it depends on modules the repo does not ship) and no reference test or
fixture holds an input/output pair for it: PARITY UNPINNED -- this file follows the cell's arithmetic line by line."""
import torch


def slice_sums(first, second, projections, p=2):
    """(n,d), (n,d), (L,d) -> (L,)  sum_i |sort(first.theta_l)_i - sort(second.theta_l)_i|^p   (:286-290)"""
    a = first.matmul(projections.transpose(0, 1)).transpose(0, 1)
    b = second.matmul(projections.transpose(0, 1)).transpose(0, 1)
    diff = torch.abs(torch.sort(a, dim=1)[0] - torch.sort(b, dim=1)[0])
    return torch.sum(torch.pow(diff, p), dim=1)


def sliced_wasserstein_distance(first, second, projections, p=2):
    """the cell's return value for given directions (:290-292)"""
    w = torch.pow(slice_sums(first, second, projections, p), 1.0 / p)
    return torch.pow(torch.pow(w, p).mean(), 1.0 / p)

"""oracle/phi_max_mirror.py -- CPU restatement of the LIVE trainer criterion
/root/reference/Point_Cloud_Resistration/losses/s2_wasserstein.py:211-262
(`max_cos_disimilarity_wassersten_distance.forward`).  TEST INFRASTRUCTURE ONLY.

That module imports POT (`import ot`, s2_wasserstein.py:8), which is not installed, so it cannot be imported to
generate fixtures and no reference fixture holds its outputs: PARITY UNPINNED -- this function follows the class
statement by statement, with the distance `CSW` injected exactly as the class takes it."""
import torch


def criterion_forward(phi, CSW, phi_op, first, second, max_iter, lam, train_or_test="train"):
    fd, sd = first.detach(), second.detach()
    trace = []
    if train_or_test == "train":
        phi.train()
        for _ in range(max_iter):                                                        # :239
            phi_op.zero_grad()                                                           # :240
            a = phi(fd)                                                                  # :241
            b = phi(sd)                                                                  # :242
            cswd = CSW(a, b)                                                             # :244
            ra = torch.sum(torch.abs(torch.linalg.vector_norm(a, dim=-1) - 1)) / (a.shape[0] * a.shape[1])   # :246
            rb = torch.sum(torch.abs(torch.linalg.vector_norm(b, dim=-1) - 1)) / (b.shape[0] * b.shape[1])   # :247
            loss = lam * (ra + rb) - cswd                                                # :248-250
            loss.backward(retain_graph=True)                                             # :251
            phi_op.step()                                                                # :252
            trace.append(float(cswd))
    else:
        phi.eval()                                                                       # :255
    a, b = phi(first), phi(second)                                                       # :257-258
    return CSW(a, b), a, b, trace                                                        # :259-260


def csw_from_pair_losses(pair_losses, p):
    """Cos_disimilarity_W.forward's batch reduction (s2_wasserstein.py:41-48) applied to per-pair distances:
    mean over the batch of distance ** (1/p) (a single pair: the value itself)."""
    if pair_losses.numel() >= 2:
        return torch.pow(pair_losses, 1.0 / p).sum() / pair_losses.numel()
    return torch.pow(pair_losses[0], 1.0 / p)

"""Losses of the geometric branch's test-time optimisation (geometric/scripts/main.py:422-456).

The reference writes the loss inline in its optimisation loop (main.py:445-451):

    loss = torch.nn.functional.mse_loss(_masks, masks_padded, reduce=False) + 100 * torch.mean(_blob['_ffd_coeffs'] ** 2)
    if image_ignores is not None:
        loss = loss * (1 - ignores_padded)
    loss = torch.mean(loss)

`silhouette_ffd_loss` is that expression as one fused HIP op (sdn_silhouette_loss_fwd / _bwd, csrc/fast_loss.hip): the loop
runs it once per iteration around a 0.9 ms frame step, where a dozen 3-9 us element-wise launches each way were 15 % of the
step.  GPU tensors only (no CPU fallback)."""
import torch


def silhouette_ffd_loss(masks, masks_target, ffd_coeffs, ignores=None):
    """masks, masks_target (and ignores) [n, 1, R, R] float32 CUDA; ffd_coeffs any shape.  Returns the scalar loss;
    differentiable wrt masks and ffd_coeffs."""
    from sdn_hip import ops
    if not masks.is_cuda:
        raise NotImplementedError('silhouette_ffd_loss runs on the GPU only (got %s)' % masks.device)
    return ops.SilhouetteLossFn.apply(masks, masks_target, ffd_coeffs, ignores)

"""TEST INFRASTRUCTURE: plain-torch stand-ins for the two HIP kernels trainer.DataParallelTrainer uses (cfm_sumsq, cfm_adam_step), so that the
bucket / accumulate / clip / all-reduce logic can be exercised on CPU tensors with the gloo backend.  Never imported by the product."""
import torch


class TorchStepKernels:
    def __init__(self):
        self.epochs = 0

    def sumsq(self, flat):
        return (flat.double() ** 2).sum().float().reshape(1)

    def adam_step(self, p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale):
        gs = g if grad_scale is None else g * grad_scale
        gs = gs + weight_decay * p
        m.mul_(betas[0]).add_(gs, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(gs, gs, value=1 - betas[1])
        bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
        p.sub_((lr / bc1) * m / (v.sqrt() / bc2 ** 0.5 + eps))

    def weights_changed(self):
        self.epochs += 1

"""EDM2 loss with DART duplication (reference edm2/loss.py:9-69): builds x = cat(images, images) + sigma*eps,
calls net(x, sigma, conditioning, just_2d=...), weights the per-frame MSE of the noised half by lambda(sigma) and
divides by the fitted mean loss."""
import math
import os
import torch
from .. import ops

FUSED = int(os.environ.get("ONIRIS_FUSED_LOSS", "1"))      # 0: the eager formulation through Precond.forward


def _core(unet):
    """The UNet behind torch's DistributedDataParallel wrapper (attribute access only: calls keep going through the wrapper)."""
    return unet.module if isinstance(unet, torch.nn.parallel.DistributedDataParallel) else unet


class EDM2Loss:
    def __init__(self, P_mean=0.5, P_std=2., sigma_data=1., context_noise_reduction=0.1):
        assert 0 <= context_noise_reduction <= 1
        self.P_mean, self.P_std, self.sigma_data = P_mean, P_std, sigma_data
        self.context_noise_reduction = context_noise_reduction

    def __call__(self, net, images, conditioning=None, sigma=None, just_2d=False, noise=None, sync=True):
        """`noise` (optional): the standard-normal draw to use (fixtures); `sync=False` skips the host round trip of the
        reference's `.cpu().item()` (:41) and returns the un-weighted loss as a device tensor.  Either way (sigma, loss) is
        logged to net.noise_weight (:43) -- on the device for HIP tensors."""
        B, T = images.shape[:2]
        assert net.training, "The model should be in training mode"
        S = 1 if just_2d else 2
        if conditioning is not None and not just_2d:
            conditioning = torch.cat((conditioning, conditioning), dim=1)
        if sigma is None:
            sigma = (torch.randn(B, T, device=images.device) * self.P_std + self.P_mean).exp()
            if not just_2d:
                ctx = torch.rand(B, 1, device=images.device).expand(-1, T) * self.context_noise_reduction
                sigma = torch.cat((ctx, sigma), dim=1)
        if noise is None:
            noise = torch.randn((B, S * T) + tuple(images.shape[2:]), dtype=images.dtype, device=images.device)
        if FUSED and self._fusable(net, images, noise, sigma):
            # same math in HIP passes (input packing, per-frame loss, its gradient, the loss tail) instead of ~45 fp32
            # elementwise launches over (B, 2T, C, H, W) / (B, T): the noised input and D_x are never materialised
            sgm = sigma.float().contiguous()
            xcl = ops.dart_input(images, noise, sgm, S, net.sigma_data)
            c_noise = sgm.log() / 4                                             # Precond.forward (networks_edm2.py:290)
            # (net.unet may be torch's DistributedDataParallel around the UNet, cs_train.py:54: the call goes THROUGH the wrapper --
            # its reducer and the inner engine for the kernel-owned weights see this forward --, out_gain is the module's own)
            Fcl, _ = net.unet.forward(xcl, c_noise, conditioning, None, False, just_2d, _cl_io=(B, S * T))
            mse = ops.dart_loss(Fcl, _core(net.unet).out_gain, images, noise, sgm, S, net.sigma_data)
            # lambda(sigma) weighting, / fitted mean loss, both means and net.noise_weight.add_data (:37-46) in one launch:
            # the (sigma, loss, position) history is appended on the device, whether or not the caller syncs
            nw = net.noise_weight
            loss, unweighted = ops.loss_tail(mse, sgm, nw.fourier_approximator.coefficients, nw.device_history(images.device),
                                             self.sigma_data)
            return loss, (unweighted.cpu().item() if sync else unweighted)
        cat_images = images if just_2d else torch.cat((images, images), dim=1)
        out, _ = net(cat_images + sigma[:, :, None, None, None] * noise, sigma, conditioning, just_2d=just_2d)
        losses = ((out[:, -T:] - images) ** 2).mean(dim=(-1, -2, -3))
        sg = sigma[:, -T:]
        losses = losses * (sg ** 2 + self.sigma_data ** 2) / (sg * self.sigma_data) ** 2
        unweighted = losses.mean().detach()
        net.noise_weight.add_data(sg, losses)                    # (device-side append for HIP tensors: no host round trip)
        if sync:
            unweighted = unweighted.cpu().item()
        mean_loss = net.noise_weight.calculate_mean_loss(sg)
        return (losses / mean_loss).mean(), unweighted

    @staticmethod
    def _fusable(net, images, noise, sigma):
        unet = _core(getattr(net, "unet", None))
        from .. import fp32 as _fp32
        if _fp32.active() or not getattr(net, "use_fp16", True):          # the fp32 path has no packed bf16 form: eager formulation
            return False
        return (images.is_cuda and images.dtype == torch.float32 and noise.dtype == torch.float32 and images.is_contiguous()
                and noise.is_contiguous() and getattr(unet, "_oniris_cl_io", False) and images.shape[2] <= 8
                and getattr(unet, "img_channels", 99) == images.shape[2] and not images.requires_grad)


def learning_rate_schedule(current_step, ref_lr=1e-2, ref_step=7e4, rampup_steps=1e3):
    """The learning rate of optimizer step `current_step` (the schedule the reference loops call, gym_train.py:110-112,
    cs_train.py:123-125; reference edm2/loss.py:63-69): `ref_lr`, divided by sqrt(step / ref_step) once the step count has
    passed `ref_step`, times a linear warm-up over the first `rampup_steps` steps.  A non-positive `ref_step` /
    `rampup_steps` switches that factor off."""
    step = float(current_step)
    past_ref = max(step / ref_step, 1.0) if ref_step > 0 else 1.0
    warm_up = min(step / rampup_steps, 1.0) if rampup_steps > 0 else 1.0
    return ref_lr / math.sqrt(past_ref) * warm_up

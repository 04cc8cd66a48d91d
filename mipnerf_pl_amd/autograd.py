"""Training path (autograd) of MipNerf.forward.  The backward kernels are not built yet; until
they are, asking for gradients fails loudly instead of silently running something else."""


def mipnerf_forward_train(model, rays, randomized, white_bkgd, t_rand=None, u_rand=None):
    raise NotImplementedError(
        "MipNerf.forward with gradients enabled: the gfx950 backward kernels (compositing / MLP dgrad+wgrad / "
        "distloss) are not implemented yet -- wrap inference in torch.no_grad().")

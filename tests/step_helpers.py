"""Shared by the -m gpu whole-step parity tests: perturbation callables that replace the model's random draws
with recorded ones (noise tensors, Bernoulli keep masks, FeatureDropout thresholds), in the two-forward and in
the forward_pair (one pass over [labelled | unlabelled]) form."""
import torch


def injected(noise, mask, u):
    """noise / mask: 5 tensors each (one per encoder level), u: 5 floats -> the list UNet_UAPS.forward(perturbations=) takes."""
    from uaps_amd import perturb
    return [lambda fs: [perturb.feature_noise_with(f, n) for f, n in zip(fs, noise)],
            lambda fs: [perturb.dropout_with(f, m) for f, m in zip(fs, mask)],
            lambda fs: [perturb.feature_dropout_with(f, uu) for f, uu in zip(fs, u)]]


def injected_pair(draws_l, draws_u):
    """draws_* = (noise, mask, u) of the labelled / unlabelled forward; every feature map is [2B, ...]."""
    from uaps_amd import perturb
    (nl, ml, ul), (nu, mu, uu) = draws_l, draws_u

    def halves(f):
        b = f.shape[0] // 2
        return f[:b].contiguous(), f[b:].contiguous()

    def both(fn, fs, dl, du):
        return [torch.cat([fn(halves(f)[0], a), fn(halves(f)[1], b)]) for f, a, b in zip(fs, dl, du)]

    return [lambda fs: both(perturb.feature_noise_with, fs, nl, nu),
            lambda fs: both(perturb.dropout_with, fs, ml, mu),
            lambda fs: [perturb.feature_dropout_with(f, (a, b)) for f, a, b in zip(fs, ul, uu)]]

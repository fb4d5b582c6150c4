"""Positional encoding with the reference's interface (models/embedder.py:6-51).

The hot path evaluates the encoding inside the HIP kernels (csrc/mlp_engine.h posenc); this torch version exists
for the small torch-side networks (RefColor, NeRF background) and for API compatibility.
"""
import torch


class Embedder:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        n_freqs = kwargs["num_freqs"]
        max_freq = kwargs["max_freq_log2"]
        if kwargs.get("log_sampling", True):
            self.freq_bands = 2.0 ** torch.linspace(0.0, max_freq, n_freqs)
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq, n_freqs)
        self.include_input = kwargs.get("include_input", True)
        # the kernel's frequencies are the octaves 2^0 .. 2^(L-1) (log sampling with max_freq_log2 = L - 1) and sin / cos
        self._log_octaves = (kwargs.get("log_sampling", True) and max_freq == n_freqs - 1
                             and list(kwargs.get("periodic_fns", [torch.sin, torch.cos])) == [torch.sin, torch.cos])
        self.periodic_fns = kwargs.get("periodic_fns", [torch.sin, torch.cos])
        self.out_dim = (d if self.include_input else 0) + d * n_freqs * len(self.periodic_fns)

    def embed(self, inputs):
        # constant 2-D float32 inputs on the GPU (the points and directions the torch-side networks of stages 2 / 3 encode):
        # one launch of fneus_embed instead of 2 L + 2 element-wise kernels
        if (inputs.is_cuda and inputs.dim() == 2 and inputs.dtype == torch.float32 and not inputs.requires_grad
                and self.include_input and self._log_octaves):
            from fneus import ops
            return ops.embed(inputs.contiguous(), len(self.freq_bands))
        outs = [inputs] if self.include_input else []
        for freq in self.freq_bands.tolist():
            for fn in self.periodic_fns:
                outs.append(fn(inputs * freq))
        return torch.cat(outs, -1)


def get_embedder(multires, input_dims=3):
    eo = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim

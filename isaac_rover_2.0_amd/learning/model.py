"""Policy / value networks that consume the rover observation layout — forward pass on the MI355X.

Mirrors ``omniisaacgymenvs/learning/model.py``: ``Layer`` (:84-121, Linear + activation), ``Encoder`` (:122-150),
``StochasticActorHeightmap.compute`` (:185-195) and ``DeterministicHeightmap.compute`` (:231-241): obs is sliced as
``[proprioceptive | sparse | dense]``, each heightmap slice goes through its encoder (default 80 → 60,
``cfg/trainSKRL/RoverPPOSKRL.yaml:7-9``), the results are concatenated with the proprioceptive values and fed to the
MLP (256 → 160 → 128, yaml :3-5) with a Tanh head of 2 (actor) or a linear head of 1 (critic).

Large batches run each encoder and the MLP + head as ONE fused kernel each (``rover_mlp_chain_forward``, f32 MFMA, activations
kept in registers between the layers); otherwise every ``Layer`` is one ``rover_linear_forward`` launch.  The slices are read in
place from ``obs_buf`` and the encoder outputs are written straight into the concat buffer, so there is no ``torch.cat``.  Training (skrl PPO,
``train.py``) stays out of scope: these classes hold plain tensors, initialised like ``nn.Linear``, and can load a
``state_dict`` saved from the reference's modules (same parameter names).
"""
from __future__ import annotations

import math

import torch


class Layer:
    def __init__(self, in_channels, out_channels, activation_function="elu", device="cuda:0", generator=None):
        bound = 1.0 / math.sqrt(in_channels) if in_channels > 0 else 0.0      # nn.Linear.reset_parameters (fan_in 0: zeros)
        self.weight = (torch.rand(out_channels, in_channels, generator=generator) * 2 - 1).mul_(bound).to(device)
        self.bias = (torch.rand(out_channels, generator=generator) * 2 - 1).mul_(bound).to(device)
        self.activation = activation_function


class HeightmapNet:
    """Shared body of the reference's two model classes; ``head_activation`` 'tanh' = actor, None = critic."""

    def __init__(self, engine, num_observations, num_sparse, num_dense, num_outputs, head_activation, mlp_features=(256, 160, 128),
                 encoder_features=(80, 60), activation_function="leakyrelu", device="cuda:0", seed=0):
        g = torch.Generator().manual_seed(seed)
        self.engine, self.device = engine, device
        self.num_sparse, self.num_dense = num_sparse, num_dense
        self.num_proprioception = num_observations - num_sparse - num_dense                # model.py:174
        mk = lambda i, o, act: Layer(i, o, act, device, g)
        self.encoder0, self.encoder1 = [], []
        i = num_sparse
        for f in encoder_features:
            self.encoder0.append(mk(i, f, activation_function)); i = f
        i = num_dense
        for f in encoder_features:
            self.encoder1.append(mk(i, f, activation_function)); i = f
        self.network = []
        i = self.num_proprioception + 2 * encoder_features[-1]                             # model.py:178
        for f in mlp_features:
            self.network.append(mk(i, f, activation_function)); i = f
        self.network.append(mk(i, num_outputs, head_activation))                          # :181-182 / :226
        # :183 — only the stochastic actor owns a log-std parameter (DeterministicHeightmap has none, :197-241)
        self.log_std_parameter = torch.zeros(num_outputs, device=device) if head_activation == "tanh" else None
        self._bufs = {}

    def _buf(self, key, rows, cols):
        b = self._bufs.get(key)
        if b is None or b.shape != (rows, cols):
            b = self._bufs[key] = torch.empty(rows, cols, device=self.device)
        return b

    def compute(self, states, fused=None):
        """model.py:185-195 / :231-241.  ``states`` [E, num_observations] float32 (may be the task's obs_buf itself).
        ``fused``: run each encoder and the MLP + head as ONE kernel each (``rover_mlp_chain_forward``: activations stay in
        registers) — default for batches of >= 20 480 rows when the layer widths fit the built tile shapes; otherwise one
        ``rover_linear_forward`` launch per layer."""
        e = states.shape[0]
        p, ns, nd = self.num_proprioception, self.num_sparse, self.num_dense
        ef = self.encoder0[-1].weight.shape[0]
        cat = self._buf("cat", e, p + 2 * ef)
        if fused is None:
            fused = True          # encoders: one fused kernel from 20 480 rows, a split-k pair below (the library decides); MLP + head: one kernel
        if (fused and ns > 0 and nd > 0 and e < 20480 and len(self.encoder0) == 2 and len(self.encoder1) == 2
                and self.engine.chain_fits(self.encoder0) and self.engine.chain_fits(self.encoder1)):
            # small batches: both encoders and the proprioception copy side by side, then the MLP + head: 3 launches instead of 6
            self.engine.chain_pair_forward(states[:, p:p + ns], self.encoder0, cat[:, p:p + ef],
                                           states[:, p + ns:p + ns + nd], self.encoder1, cat[:, p + ef:p + 2 * ef],
                                           copy_src=states, copy_dst=cat, copy_cols=p)
            if self.engine.chain_fits(self.network):
                return self.engine.chain_forward(cat, self.network, self._buf(("mlp", len(self.network) - 1), e, self.network[-1].weight.shape[0]))
            x = cat
            for li, layer in enumerate(self.network):
                x = self.engine.linear_forward(x, layer.weight, layer.bias, layer.activation, self._buf(("mlp", li), e, layer.weight.shape[0]))
            return x
        cat[:, 0:p] = states[:, 0:p]
        for enc, lo, n, col in ((self.encoder0, p, ns, p), (self.encoder1, p + ns, nd, p + ef)):
            x = states[:, lo:lo + n]
            if fused and n > 0 and self.engine.chain_fits(enc):
                self.engine.chain_forward(x, enc, cat[:, col:col + ef])
                continue
            for li, layer in enumerate(enc):
                last = li == len(enc) - 1
                out = cat[:, col:col + ef] if last else self._buf(("enc", col, li), e, layer.weight.shape[0])
                x = self.engine.linear_forward(x, layer.weight, layer.bias, layer.activation, out)
        if fused and self.engine.chain_fits(self.network):
            return self.engine.chain_forward(cat, self.network, self._buf(("mlp", len(self.network) - 1), e, self.network[-1].weight.shape[0]))
        x = cat
        for li, layer in enumerate(self.network):
            x = self.engine.linear_forward(x, layer.weight, layer.bias, layer.activation, self._buf(("mlp", li), e, layer.weight.shape[0]))
        return x

    # ---- interop with the reference's nn.Module parameter names --------------------------------------------
    def state_dict(self):
        sd = {} if self.log_std_parameter is None else {"log_std_parameter": self.log_std_parameter}
        for name, enc in (("encoder0", self.encoder0), ("encoder1", self.encoder1)):
            for i, l in enumerate(enc):
                sd[f"{name}.encoder.{i}.layer.0.weight"] = l.weight
                sd[f"{name}.encoder.{i}.layer.0.bias"] = l.bias
        for i, l in enumerate(self.network[:-1]):
            sd[f"network.{i}.layer.0.weight"] = l.weight
            sd[f"network.{i}.layer.0.bias"] = l.bias
        k = len(self.network) - 1
        sd[f"network.{k}.weight"] = self.network[-1].weight
        sd[f"network.{k}.bias"] = self.network[-1].bias
        return sd

    def load_state_dict(self, sd):
        for k, v in self.state_dict().items():
            if k in sd:
                v.copy_(sd[k].to(self.device))


def StochasticActorHeightmap(engine, task, **kw):
    hm = task.Camera.heightmap
    return HeightmapNet(engine, task.num_observations, hm.get_num_sparse_vector(), hm.get_num_dense_vector(), task.num_actions,
                        "tanh", device=task.device, **kw)


def DeterministicHeightmap(engine, task, **kw):
    hm = task.Camera.heightmap
    return HeightmapNet(engine, task.num_observations, hm.get_num_sparse_vector(), hm.get_num_dense_vector(), 1, None,
                        device=task.device, **kw)

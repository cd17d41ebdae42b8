"""Host-side mirror of the reference's module surface for the Glow hot path.

Same class names, constructor kwargs, attribute names and ``state_dict`` keys/shapes as
``models/layers.py``, ``models/flows.py`` and ``models/kd_flows.py`` of the reference, so that
``pl_module.py`` can use these classes unchanged -- but the modules only
*hold* parameters: all arithmetic is done by the hand-written HIP kernels in ``csrc/`` through
the C ABI (``include/nfk.h``).  There is no CPU path; tensors must live on an MI355X.

Activations are NHWC in HBM.  Tensors handed back to the caller have the reference's logical
shape ``[B, C, H, W]`` with channels-last strides (``x.permute(0, 3, 1, 2)`` views).
"""
import ctypes as C
import logging
import math
import os

import torch
import torch.nn as nn

from . import capi
from .capi import PrepDesc, PrepGrad, RepackDesc, Split2d as CSplit2d, Step2d, Step2dGrads, check, ptr

logger = logging.getLogger(__name__)


def _a4(n):
    return (n + 3) & ~3


# ======================================================================================
# parameter-holding modules (names/shapes follow the reference)
# ======================================================================================
class _ActNorm(nn.Module):
    """models/layers.py:64-142.  bias/logs are [1,C,1,1] (2-D) or [1,C] (1-D)."""

    def __init__(self, num_features, scale=1.0, is_1d=False):
        super().__init__()
        self.is_1d = is_1d
        size = [1, num_features] + ([] if is_1d else [1, 1])
        self.bias = nn.Parameter(torch.zeros(*size))
        self.logs = nn.Parameter(torch.zeros(*size))
        self.num_features = num_features
        self.scale = scale
        self.inited = False

    def initialize_parameters(self, input):
        """Data-dependent init (models/layers.py:83-99); one-off, so plain tensor ops."""
        if not self.training:
            raise ValueError("In Eval mode, but ActNorm not inited")
        dims = [0] + ([] if self.is_1d else [2, 3])
        with torch.no_grad():
            bias = -torch.mean(input, dim=dims, keepdim=True)
            var = torch.mean((input + bias) ** 2, dim=dims, keepdim=True)
            # in-place on the Parameters themselves (not on .data): bumps their version counters, which is what tells a
            # FlowEngine that cached W.diag(e^logs) products of a frozen model are stale
            self.bias.copy_(bias)
            self.logs.copy_(torch.log(self.scale / (torch.sqrt(var) + 1e-6)))
            self.inited = True

    def forward(self, input, logdet=None, reverse=False):
        self._check_input_dim(input)
        if not self.inited:
            self.initialize_parameters(input)
        from .functional import actnorm_apply
        return actnorm_apply(self, input, logdet, reverse)


class ActNorm2d(_ActNorm):
    def __init__(self, num_features, scale=1.0):
        super().__init__(num_features, scale, is_1d=False)

    def _check_input_dim(self, input):
        assert len(input.size()) == 4
        assert input.size(1) == self.num_features, (
            "[ActNorm]: input should be in shape as `B x C x H x W`,"
            " channels should be {} rather than {}".format(self.num_features, input.size()))


class ActNorm1d(_ActNorm):
    def __init__(self, num_features, scale=1.0):
        super().__init__(num_features, scale, is_1d=True)

    def _check_input_dim(self, input):
        assert len(input.size()) == 2
        assert input.size(1) == self.num_features, (
            "[ActNorm]: input should be in shape as `B x C`,"
            " channels should be {} rather than {}".format(self.num_features, input.size()))


class LinearZeros(nn.Module):
    """models/layers.py:173-187 (used only for learn_top / y_condition heads: [B, classes]-sized
    work, done with tensor ops on the device)."""

    def __init__(self, in_channels, out_channels, logscale_factor=3):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels)
        self.linear.weight.data.zero_()
        self.linear.bias.data.zero_()
        self.logscale_factor = logscale_factor
        self.logs = nn.Parameter(torch.zeros(out_channels))

    def forward(self, input):
        return self.linear(input) * torch.exp(self.logs * self.logscale_factor)


class Conv2d(nn.Module):
    """models/layers.py:190-228: SAME conv without bias + ActNorm2d.  Parameter holder."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding="same",
                 do_actnorm=True, weight_std=0.05):
        super().__init__()
        if tuple(stride) != (1, 1) or padding != "same" or not do_actnorm:
            raise NotImplementedError("the hot path only uses stride-1 SAME convs with ActNorm")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, bias=False)
        torch.nn.init.xavier_normal_(self.conv.weight)
        self.actnorm = ActNorm2d(out_channels)
        self.do_actnorm = do_actnorm


class Conv2dZeros(nn.Module):
    """models/layers.py:231-260: SAME conv with bias, times exp(3*logs).  Parameter holder."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding="same",
                 logscale_factor=3):
        super().__init__()
        if tuple(stride) != (1, 1) or padding != "same" or logscale_factor != 3:
            raise NotImplementedError("the hot path only uses stride-1 SAME convs, logscale 3")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride)
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()
        self.logscale_factor = logscale_factor
        self.logs = nn.Parameter(torch.zeros(out_channels, 1, 1))


class Permute2d(nn.Module):
    """models/layers.py:263-290: fixed channel gather (indices are plain attributes upstream)."""

    def __init__(self, num_channels, shuffle):
        super().__init__()
        self.num_channels = num_channels
        self.indices = torch.arange(num_channels - 1, -1, -1, dtype=torch.long)
        if shuffle:
            self.reset_indices()

    def reset_indices(self):
        self.indices = self.indices[torch.randperm(self.indices.shape[0])]

    @property
    def indices_inverse(self):
        inv = torch.empty_like(self.indices)
        inv[self.indices] = torch.arange(self.num_channels)
        return inv


class InvertibleConv1x1(nn.Module):
    """models/layers.py:330-421.  Same init (QR of a Gaussian, then LU) and the same
    parameters/buffers: lower, log_s, upper (params), p, sign_s (buffers) or weight."""

    def __init__(self, num_channels, LU_decomposed, is_1d=False):
        super().__init__()
        self.is_1d = is_1d
        w_shape = [num_channels, num_channels]
        w_init = torch.linalg.qr(torch.randn(*w_shape))[0].contiguous()
        if not LU_decomposed:
            self.weight = nn.Parameter(w_init.clone())
        else:
            p, lower, upper = torch.linalg.lu(w_init)
            s = torch.diag(upper)
            self.register_buffer("p", p.contiguous())
            self.register_buffer("sign_s", torch.sign(s).contiguous())
            self.lower = nn.Parameter(lower.contiguous())
            self.log_s = nn.Parameter(torch.log(torch.abs(s)).contiguous())
            self.upper = nn.Parameter(torch.triu(upper, 1).contiguous())
        self.w_shape = w_shape
        self.LU_decomposed = LU_decomposed

    def forward(self, input, logdet=None, reverse=False):
        from .functional import invconv_apply
        return invconv_apply(self, input, logdet, reverse)


def get_block_2d(in_channels, out_channels, hidden_channels):
    """models/flows.py:25-34"""
    return nn.Sequential(
        Conv2d(in_channels, hidden_channels),
        nn.ReLU(inplace=False),
        Conv2d(hidden_channels, hidden_channels, kernel_size=(1, 1)),
        nn.ReLU(inplace=False),
        Conv2dZeros(hidden_channels, out_channels),
    )


def get_block_1d(in_features, out_features, hidden_features):
    """models/flows.py:37-52"""
    return nn.Sequential(
        nn.Linear(in_features, hidden_features), nn.ReLU(inplace=False),
        nn.Linear(hidden_features, hidden_features), nn.ReLU(inplace=False),
        nn.Linear(hidden_features, hidden_features), nn.ReLU(inplace=False),
        nn.Linear(hidden_features, hidden_features), nn.ReLU(inplace=False),
        nn.Linear(hidden_features, hidden_features), nn.Tanh(),
        nn.Linear(hidden_features, out_features),
    )


class SqueezeLayer(nn.Module):
    """models/layers.py:316-327"""

    def __init__(self, factor):
        super().__init__()
        if factor != 2:
            raise NotImplementedError("only factor-2 squeeze is on the hot path")
        self.factor = factor

    def forward(self, input, logdet=None, reverse=False, **kwargs):
        from .engine import run_single_layer
        return run_single_layer(self, input, None, logdet, reverse)


class Split2d(nn.Module):
    """models/layers.py:293-313"""

    def __init__(self, num_channels):
        super().__init__()
        self.conv = Conv2dZeros(num_channels // 2, num_channels)

    def forward(self, input, logdet=0.0, reverse=False, temperature=None, **kwargs):
        from .engine import run_single_layer
        return run_single_layer(self, input, None, logdet, reverse, temperature=temperature,
                                eps=kwargs.get("eps"))


class FlowStep(nn.Module):
    """models/flows.py:55-202: actnorm -> permutation -> affine/additive coupling."""

    def __init__(self, in_channels, hidden_channels, actnorm_scale, flow_permutation, flow_coupling,
                 LU_decomposed, is_1d=False, condition_features=0):
        super().__init__()
        self.is_1d = is_1d
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.flow_coupling = flow_coupling
        self.flow_permutation_type = flow_permutation
        self.condition_features = condition_features
        self.actnorm = (ActNorm1d if is_1d else ActNorm2d)(in_channels, actnorm_scale)
        if flow_permutation == "invconv":
            self.invconv = InvertibleConv1x1(in_channels, LU_decomposed=LU_decomposed, is_1d=is_1d)
        elif flow_permutation == "shuffle":
            if is_1d:
                raise RuntimeError("Permutation is not supported is 1d mode")
            self.shuffle = Permute2d(in_channels, shuffle=True)
        else:
            if is_1d:
                raise RuntimeError("Permutation is not supported is 1d mode")
            self.reverse = Permute2d(in_channels, shuffle=False)
        if flow_coupling == "additive":
            out_block = in_channels - in_channels // 2
        elif flow_coupling == "affine":
            out_block = (in_channels - in_channels // 2) * 2
        else:
            raise NameError(f"Unknown coupling type: {flow_coupling}")
        in_block = in_channels // 2 + condition_features
        self.block = (get_block_1d if is_1d else get_block_2d)(in_block, out_block, hidden_channels)

    def forward(self, input, y_onehot=None, logdet=None, reverse=False):
        from .engine import run_single_layer
        return run_single_layer(self, input, y_onehot, logdet, reverse)

    def normal_flow(self, input, y_onehot, logdet):
        return self.forward(input, y_onehot, logdet, False)

    def reverse_flow(self, input, y_onehot, logdet):
        return self.forward(input, y_onehot, logdet, True)


def made_masks(d, hidden, ncond=0):
    """Autoregressive masks of a MADE whose MLP has the shape of get_block_1d (Germain et al. 2015; MAF, Papamakarios et
    al. 2017): input degrees 1..d (condition features 0: visible to everything), hidden degrees cycling over 1..d-1,
    hidden->hidden allowed when degree_out >= degree_in, output (mu_i, alpha_i interleaved) allowed when i+1 > degree_in."""
    deg_in = torch.cat([torch.arange(1, d + 1), torch.zeros(ncond, dtype=torch.long)])
    deg_h = torch.arange(hidden) % max(d - 1, 1) + 1
    deg_out = torch.arange(1, d + 1).repeat_interleave(2)
    masks = [(deg_h[:, None] >= deg_in[None, :]).float()]
    masks += [(deg_h[:, None] >= deg_h[None, :]).float() for _ in range(4)]
    masks.append((deg_out[:, None] > deg_h[None, :]).float())
    return masks


class MADEStep(nn.Module):
    """One block of a masked autoregressive flow -- BASELINE.json / north_star name "MAF MADE blocks"; the reference
    repository has NO such code (its README mentions MAF once), so this follows the published
    algorithm (Papamakarios et al. 2017, eq. 3-5) in the reference's own building blocks and is PARITY UNPINNED:
        ActNorm1d -> reverse permutation -> u_i = (y_i - mu_i(y_<i)) * exp(-alpha_i(y_<i)),  log|det| = -sum alpha_i
    with (mu, alpha) from a MADE: the get_block_1d MLP (models/flows.py:37-52) whose weights are multiplied with
    autoregressive masks.  Runs as ONE fused launch per direction through the tabular kernels (nfk_tab_step.made)."""

    is_1d = True
    flow_coupling = "made"
    flow_permutation_type = "reverse"

    def __init__(self, in_channels, hidden_channels, actnorm_scale=1.0, condition_features=0):
        super().__init__()
        if in_channels < 2:
            raise ValueError("a MADE block needs at least 2 features")
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.condition_features = condition_features
        self.actnorm = ActNorm1d(in_channels, actnorm_scale)
        self.reverse = Permute2d(in_channels, shuffle=False)
        self.block = get_block_1d(in_channels + condition_features, 2 * in_channels, hidden_channels)
        for k, m in enumerate(made_masks(in_channels, hidden_channels, condition_features)):
            self.register_buffer(f"mask{k}", m)

    def masks(self):
        return [getattr(self, f"mask{k}") for k in range(6)]

    def forward(self, input, y_onehot=None, logdet=None, reverse=False):
        from .engine import run_single_layer
        return run_single_layer(self, input, y_onehot, logdet, reverse)


class FlowNet(nn.Module):
    """models/flows.py:205-295: L x (squeeze, K steps, split)."""

    def __init__(self, image_shape, hidden_channels, K, L, actnorm_scale, flow_permutation, flow_coupling,
                 LU_decomposed, is_1d=False, condition_features=0):
        super().__init__()
        self.is_1d = is_1d
        self.layers = nn.ModuleList()
        self.output_shapes = []
        self.K, self.L = K, L
        if not is_1d:
            H, W, Cc = image_shape
        else:
            Cc = image_shape[0]
        for i in range(L):
            if not is_1d:
                Cc, H, W = Cc * 4, H // 2, W // 2
                self.layers.append(SqueezeLayer(factor=2))
                self.output_shapes.append([-1, Cc, H, W])
            for _ in range(K):
                if flow_coupling == "made":   # MAF: a stack of MADE blocks (extension: no upstream counterpart)
                    if not is_1d:
                        raise NameError("flow_coupling='made' is a 1-D (tabular) block")
                    self.layers.append(MADEStep(Cc, hidden_channels, actnorm_scale, condition_features))
                    self.output_shapes.append([-1, Cc])
                    continue
                self.layers.append(FlowStep(in_channels=Cc, hidden_channels=hidden_channels,
                                            actnorm_scale=actnorm_scale, flow_permutation=flow_permutation,
                                            flow_coupling=flow_coupling, LU_decomposed=LU_decomposed,
                                            is_1d=is_1d, condition_features=condition_features))
                self.output_shapes.append([-1, Cc] if is_1d else [-1, Cc, H, W])
            if i < L - 1 and not is_1d:
                self.layers.append(Split2d(num_channels=Cc))
                self.output_shapes.append([-1, Cc // 2, H, W])
                Cc = Cc // 2
        self._engine = None

    def forward(self, input, y_onehot=None, logdet=0.0, reverse=False, temperature=None):
        if reverse:
            return self.decode(input, y_onehot=y_onehot, temperature=temperature)
        return self.encode(input, y_onehot=y_onehot, logdet=logdet)

    def engine(self):
        from .engine import FlowEngine
        if self._engine is None:
            self._engine = FlowEngine(list(self.layers))
        return self._engine

    def _encode_all(self, z, y_onehot, logdet, noise=None, keep=None, hints_only=False):
        return self.engine().encode(z, y_onehot, logdet, noise, keep, hints_only)

    def _decode_all(self, z, y_onehot, temperature, eps=None):
        return self.engine().decode(z, y_onehot, temperature, eps)

    def encode(self, z, y_onehot=None, logdet=0.0):
        outs, logdet = self._encode_all(z, y_onehot, logdet)
        return outs[-1], logdet

    def decode(self, z, y_onehot=None, temperature=None):
        return self._decode_all(z, y_onehot, temperature)[-1]


class FlowNetGetAllOutputs(FlowNet):
    """models/kd_flows.py:15-73: encode/decode return the list of every layer's output."""

    def encode(self, z, y_onehot=None, logdet=0.0):
        return self._encode_all(z, y_onehot, logdet)

    def decode(self, z, y_onehot=None, temperature=None):
        return self._decode_all(z, y_onehot, temperature)


def split_feature(tensor, type="split"):
    """models/utils.py:44-52"""
    Cc = tensor.size(1)
    if type == "split":
        return tensor[:, : Cc // 2, ...], tensor[:, Cc // 2:, ...]
    return tensor[:, 0::2, ...], tensor[:, 1::2, ...]


def gaussian_sample(mean, logs, temperature=1):
    """models/layers.py:26-29"""
    return torch.normal(mean, torch.exp(logs) * temperature)


class Glow(nn.Module):
    """models/flows.py:298-438"""

    _flow_cls = FlowNet

    def __init__(self, image_shape, hidden_channels, K, L, actnorm_scale, flow_permutation, flow_coupling,
                 LU_decomposed, y_classes, learn_top, y_condition, is_1d=False):
        super().__init__()
        self.flow = self._flow_cls(image_shape=image_shape, hidden_channels=hidden_channels, K=K, L=L,
                                   actnorm_scale=actnorm_scale, flow_permutation=flow_permutation,
                                   flow_coupling=flow_coupling, LU_decomposed=LU_decomposed, is_1d=is_1d,
                                   condition_features=y_classes if y_condition else 0)
        self.is_1d = is_1d
        self.y_classes = y_classes
        self.y_condition = y_condition
        self.learn_top = learn_top
        Ct = self.flow.output_shapes[-1][1]
        if learn_top:
            self.learn_top_fn = LinearZeros(Ct * 2, Ct * 2) if is_1d else Conv2dZeros(Ct * 2, Ct * 2)
        if y_condition:
            self.project_ycond = LinearZeros(y_classes, 2 * Ct)
            self.project_class = LinearZeros(Ct, y_classes)
        self.register_buffer(
            "prior_h", torch.zeros([1, Ct * 2] + ([] if is_1d else list(self.flow.output_shapes[-1][2:]))))

    # -- prior (models/flows.py:367-391): [B or 32, 2C, ...]-sized, parameter-only work ------------
    def _prior_tensor(self, batch, y_onehot):
        h = self.prior_h
        if self.learn_top:
            if self.is_1d:
                h = self.learn_top_fn(h)
            else:
                from .functional import conv2d_zeros_small
                h = conv2d_zeros_small(self.learn_top_fn, h)
        if self.y_condition:
            assert y_onehot is not None
            yp = self.project_ycond(y_onehot)
            h = h + yp.view([yp.shape[0], h.shape[1]] + ([] if self.is_1d else [1, 1]))
        if h.shape[0] == 1 and batch != 1:
            h = h.expand(batch, *h.shape[1:])
        return h

    def prior(self, data, y_onehot=None):
        if data is not None:
            batch = data.shape[0]
        else:
            batch = y_onehot.size(0) if y_onehot is not None else 32  # hard-coded upstream
        h = self._prior_tensor(batch, y_onehot)
        return split_feature(h, "split")

    def forward(self, x=None, y_onehot=None, z=None, temperature=None, reverse=False, noise=None, eps=None, keep=None,
                hints_only=False):
        """The reference's signature (models/flows.py:393-397) plus three optional extensions: `noise` / `eps` supply the
        dequantisation / Split2d draws, `keep` (no-grad passes) names the layer outputs the caller will read -- the KD hint
        indices of pl_module.py:262-282 -- so that the others are neither kept nor, where a level runs as one launch,
        written to HBM at all (their entries in the returned list are None; the last layer is always there).  `hints_only`
        (with `keep`, image flows, no gradient): the caller wants those outputs and nothing else -- the returned bits/dim
        is None, and what only the likelihood needs (Split2d's prior convolution, the log-det sums, the prior) is not run."""
        if reverse:
            return self.reverse_flow(z, y_onehot, temperature, eps=eps)
        if keep is not None:
            if (hints_only and not self.is_1d and not self.y_condition and not torch.is_grad_enabled()
                    and os.environ.get("NFK_HINTS_ONLY", "1") != "0"):
                return self._forward_all(x, y_onehot, noise, keep, hints_only=True)
            return self.normal_flow(x, y_onehot, noise=noise, keep=keep)
        return self.normal_flow(x, y_onehot, noise=noise)

    def _forward_all(self, x, y_onehot, noise, keep=None, hints_only=False):
        """Shared by Glow / GlowGetAllOutputs: returns (all outputs, bpd, y_logits)."""
        from .engine import prior_bpd
        capi.require_cuda(x, "x")
        if self.is_1d:
            b, c = x.shape
            logdet = torch.zeros(b, device=x.device, dtype=torch.float32)
            dims = 1
        else:
            b, c, h, w = x.shape
            # uniform_binning_correction (models/utils.py:26-41): x += U(0,1/256) IN PLACE; the add is
            # folded into the first squeeze kernel.  `noise` lets a caller supply the draw.
            if noise is None:
                noise = torch.empty_like(x).uniform_(0, 1.0 / 256)
            logdet = torch.full((b,), -math.log(256.0) * c * h * w, device=x.device, dtype=torch.float32)
            dims = c * h * w
        if hints_only:
            outs, _ = self.flow._encode_all(x, y_onehot, logdet, noise, keep, hints_only=True)
            return outs, None, None
        outs, objective_ld = self.flow._encode_all(x, y_onehot, logdet, noise, keep)
        last = outs[-1]
        prior_t = self._prior_tensor(b, y_onehot)
        scale = 1.0 if self.is_1d else 1.0 / (math.log(2.0) * dims)
        bpd = prior_bpd(last, prior_t, objective_ld, scale)
        y_logits = None
        if self.y_condition:
            pooled = last if self.is_1d else last.mean(dim=[2, 3])
            y_logits = self.project_class(pooled)
        return outs, bpd, y_logits

    def normal_flow(self, x, y_onehot=None, noise=None):
        outs, bpd, y_logits = self._forward_all(x, y_onehot, noise)
        return outs[-1], bpd, y_logits

    def reverse_flow(self, z, y_onehot, temperature, eps=None):
        if z is None:
            mean, logs = self.prior(z, y_onehot)
            z = gaussian_sample(mean, logs, temperature)
        outs = self.flow._decode_all(z, y_onehot, temperature, eps)
        return outs if isinstance(self.flow, FlowNetGetAllOutputs) else outs[-1]

    def set_compute_dtype(self, name):
        """"f32": the reference's precision (parity mode).  "bf16": bf16 operands with fp32 accumulation
        in the coupling-net convolutions (throughput mode; everything else stays fp32)."""
        self.flow.engine().set_compute_dtype(name)
        return self

    def set_fused(self, on=True):
        """bf16 mode only: run each FlowStep forward as one fused launch (default) or as five kernels."""
        self.flow.engine().fused = bool(on)
        return self

    def set_actnorm_init(self):
        for _, module in self.named_modules():
            if isinstance(module, (ActNorm2d, ActNorm1d)):
                module.inited = True


class GlowGetAllOutputs(Glow):
    """models/kd_flows.py:76-152"""

    _flow_cls = FlowNetGetAllOutputs

    def normal_flow(self, x, y_onehot=None, noise=None, keep=None):
        return self._forward_all(x, y_onehot, noise, keep)


def create_glow_model(config):
    """models/kd_flows.py:155-159"""
    model = GlowGetAllOutputs(**config)
    model.set_actnorm_init()
    return model


def inherit_permutation_matrix(student, teacher, student_kd_indices, teacher_kd_indices):
    """models/kd_flows.py:162-179: student's permutation at a hint layer = product of the teacher's
    permutations since the previous hint layer."""
    k = 0
    acc = None
    for t_id, t_layer in enumerate(teacher.flow.layers):
        if t_id == teacher_kd_indices[k]:
            # `None @ p` upstream when no FlowStep lies between two hint layers: the same TypeError here
            student.flow.layers[student_kd_indices[k]].invconv.p = acc @ t_layer.invconv.p
            k += 1
            acc = None
        elif isinstance(t_layer, FlowStep):
            acc = t_layer.invconv.p if acc is None else acc @ t_layer.invconv.p
    eng = getattr(student.flow, "_engine", None)
    if eng is not None:
        eng.mark_dirty()   # cached W = P.L.U products are stale

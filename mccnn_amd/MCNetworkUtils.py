"""Dense helper layers used between the MC convolutions: the counterpart of the reference's
utils/MCNetworkUtils.py with the same function names, argument orders, variable names and initialisers, on torch.

The reference creates variables through TF's global variable scope (tf.get_variable); here they live in a
VariableStore (the ConvolutionBuilder keeps its own in the same way). Batch normalisation follows
tf.layers.batch_normalization's defaults (momentum 0.99, epsilon 1e-3, gamma/beta) and -- because the batch is sharded
cloud-per-GPU while the reference normalises over the WHOLE batch (utils/MCNetworkUtils.py:140) -- synchronises its
statistics across ranks with one small all-reduce when torch.distributed is initialised (point counts differ per rank,
so sums and counts are reduced, not means).
"""
import math

import torch
import torch.distributed as dist


class VariableStore(torch.nn.Module):
    """name -> torch.nn.Parameter / buffer; tf.get_variable + tf.add_to_collection semantics on a torch.nn.Module, so
    that `parameters()`, `state_dict()`, `.to()`, optimisers and DDP wrappers see the variables under the reference's
    names (`Reduce_1_weights`, `Final_Logits_BN_h1/gamma`, ...). Variables are created on first use (tf.get_variable),
    so load_state_dict() also accepts names that do not exist yet."""

    def __init__(self, device=None):
        super().__init__()
        self.device = device
        self.collections_ = {}

    @property
    def variables_(self):
        return self._parameters

    @property
    def buffers_(self):
        return self._buffers

    def get_variable(self, name, shape, init, device=None):
        p = self._parameters.get(name)
        if p is None:
            p = torch.nn.Parameter(init(torch.empty(shape, dtype=torch.float32, device=device or self.device)))
            self.register_parameter(name, p)
        elif tuple(p.shape) != tuple(shape):
            raise RuntimeError("variable %s exists with shape %s, requested %s" % (name, tuple(p.shape), shape))
        return p

    def get_buffer(self, name, shape, fill, device=None):
        b = self._buffers.get(name)
        if b is None:
            b = torch.full(shape, float(fill), dtype=torch.float32, device=device or self.device)
            self.register_buffer(name, b)
        return b

    def add_to_collection(self, coll, p):
        lst = self.collections_.setdefault(coll, [])
        if not any(q is p for q in lst):
            lst.append(p)

    def get_collection(self, coll):
        return list(self.collections_.get(coll, []))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # lazily created variables: adopt what is not there yet (also when a parent module's load_state_dict() recurses
        # into this one -- variable names hold no '.', so every key below `prefix` without one is this module's own)
        for k, v in state_dict.items():
            name = k[len(prefix):] if k.startswith(prefix) else None
            if name and "." not in name and name not in self._parameters and name not in self._buffers:
                if name.endswith("/moving_mean") or name.endswith("/moving_variance"):
                    self.register_buffer(name, v.detach().clone())
                else:
                    self.register_parameter(name, torch.nn.Parameter(v.detach().clone()))
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


_DEFAULT_STORE = VariableStore()


def get_default_store():
    return _DEFAULT_STORE


def reset_default_store(device=None):
    global _DEFAULT_STORE
    _DEFAULT_STORE = VariableStore(device)
    return _DEFAULT_STORE


def _fan_avg_uniform(fan_in, fan_out):
    limit = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))  # variance_scaling_initializer(1.0, 'FAN_AVG', uniform=True)

    def init(t):
        with torch.no_grad():
            t.uniform_(-limit, limit)
        return t
    return init


def _glorot_uniform(fan_in, fan_out):
    # tf.get_variable's default initializer (used for ..._weights2 in MLP_2_hidden, MCNetworkUtils.py:55)
    limit = math.sqrt(6.0 / (fan_in + fan_out))

    def init(t):
        with torch.no_grad():
            t.uniform_(-limit, limit)
        return t
    return init


def _zeros(t):
    with torch.no_grad():
        t.zero_()
    return t


def _ones(t):
    with torch.no_grad():
        t.fill_(1.0)
    return t


def batch_normalization(inputs, training, name, store=None, momentum=0.99, epsilon=1e-3, sync=True):
    """tf.layers.batch_normalization(inputs, training=..., name=...) over axis 0 of an [n, c] tensor."""
    st = store or _DEFAULT_STORE
    c = inputs.shape[1]
    dev = inputs.device
    gamma = st.get_variable(name + "/gamma", (c,), _ones, dev)
    beta = st.get_variable(name + "/beta", (c,), _zeros, dev)
    mov_mean = st.get_buffer(name + "/moving_mean", (c,), 0.0, dev)
    mov_var = st.get_buffer(name + "/moving_variance", (c,), 1.0, dev)
    distributed = sync and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if training and not distributed and inputs.shape[0] > 1:
        # single process: the fused batch-norm kernels (one forward, one backward launch instead of ~30 element-wise
        # ones; the network is host-bound). Same arithmetic: biased batch variance for the normalisation AND for the
        # moving average, as tf.layers.batch_normalization does (torch's own running_var update would be unbiased).
        with torch.no_grad():
            var, mean = torch.var_mean(inputs, 0, unbiased=False)
            mov_mean.mul_(momentum).add_(mean, alpha=1.0 - momentum)
            mov_var.mul_(momentum).add_(var, alpha=1.0 - momentum)
        return torch.nn.functional.batch_norm(inputs, None, None, gamma, beta, True, 0.0, epsilon)
    if training:
        n = torch.tensor([float(inputs.shape[0])], dtype=torch.float32, device=dev)
        s1 = inputs.sum(0)
        s2 = (inputs * inputs).sum(0)
        if distributed:
            import torch.distributed.nn.functional as dfn
            packed = dfn.all_reduce(torch.cat([s1, s2, n]), op=dist.ReduceOp.SUM)
            s1, s2, n = packed[:c], packed[c:2 * c], packed[2 * c:]
        mean = s1 / n
        var = (s2 / n - mean * mean).clamp_min(0.0)
        with torch.no_grad():
            mov_mean.mul_(momentum).add_(mean.detach(), alpha=1.0 - momentum)
            mov_var.mul_(momentum).add_(var.detach(), alpha=1.0 - momentum)
    else:
        mean, var = mov_mean, mov_var
    return (inputs - mean) * torch.rsqrt(var + epsilon) * gamma + beta


def _dropout(x, keepProb, training=True):
    if keepProb is None or keepProb is False:
        return x
    return torch.nn.functional.dropout(x, p=1.0 - float(keepProb), training=training)


def MLP_2_hidden(features, numInputFeatures, hidden1_units, hidden2_units, numOutFeatures, layerName, keepProb,
                 isTraining, useDropOut=False, useInitBN=True, store=None):
    """utils/MCNetworkUtils.py:20-69."""
    st = store or _DEFAULT_STORE
    dev = features.device
    if useInitBN:
        features = batch_normalization(features, isTraining, layerName + "_BN_Init", st)
    w = st.get_variable(layerName + '_weights1', (numInputFeatures, hidden1_units),
                        _fan_avg_uniform(numInputFeatures, hidden1_units), dev)
    st.add_to_collection('weight_decay_loss', w)
    b = st.get_variable(layerName + '_biases1', (hidden1_units,), _zeros, dev)
    hidden1 = torch.relu(batch_normalization(features @ w + b, isTraining, layerName + "_BN_h1", st))
    if useDropOut:
        hidden1 = _dropout(hidden1, keepProb, isTraining)
    w = st.get_variable(layerName + '_weights2', (hidden1_units, hidden2_units),
                        _glorot_uniform(hidden1_units, hidden2_units), dev)
    st.add_to_collection('weight_decay_loss', w)
    b = st.get_variable(layerName + '_biases2', (hidden2_units,), _zeros, dev)
    hidden2 = torch.relu(batch_normalization(hidden1 @ w + b, isTraining, layerName + "_BN_h2", st))
    if useDropOut:
        hidden2 = _dropout(hidden2, keepProb, isTraining)
    w = st.get_variable(layerName + '_weights3', (hidden2_units, numOutFeatures),
                        _fan_avg_uniform(hidden2_units, numOutFeatures), dev)
    st.add_to_collection('weight_decay_loss', w)
    b = st.get_variable(layerName + '_biases3', (numOutFeatures,), _zeros, dev)
    return hidden2 @ w + b


def MLP_1_hidden(features, numInputFeatures, hidden_units, numOutFeatures, layerName, keepProb, isTraining,
                 useDropOut=False, store=None):
    """utils/MCNetworkUtils.py:72-106."""
    st = store or _DEFAULT_STORE
    dev = features.device
    w = st.get_variable(layerName + '_weights1', (numInputFeatures, hidden_units),
                        _fan_avg_uniform(numInputFeatures, hidden_units), dev)
    st.add_to_collection('weight_decay_loss', w)
    b = st.get_variable(layerName + '_biases1', (hidden_units,), _zeros, dev)
    hidden = torch.relu(batch_normalization(features @ w + b, isTraining, layerName + "_BN_h", st))
    if useDropOut:
        hidden = _dropout(hidden, keepProb, isTraining)
    w = st.get_variable(layerName + '_weights2', (hidden_units, numOutFeatures),
                        _fan_avg_uniform(hidden_units, numOutFeatures), dev)
    st.add_to_collection('weight_decay_loss', w)
    b = st.get_variable(layerName + '_biases2', (numOutFeatures,), _zeros, dev)
    return hidden @ w + b


def conv_1x1(layerName, inputs, numInputs, numOutFeatures, store=None):
    """utils/MCNetworkUtils.py:109-126."""
    st = store or _DEFAULT_STORE
    dev = inputs.device
    w = st.get_variable(layerName + '_weights', (numInputs, numOutFeatures), _fan_avg_uniform(numInputs, numOutFeatures), dev)
    st.add_to_collection('weight_decay_loss', w)
    b = st.get_variable(layerName + '_biases', (numOutFeatures,), _zeros, dev)
    return inputs @ w + b


def batch_norm_RELU_drop_out(layerName, inFeatures, isTraining, usedDropOut, keepProb, store=None):
    """utils/MCNetworkUtils.py:129-144."""
    st = store or _DEFAULT_STORE
    x = torch.relu(batch_normalization(inFeatures, isTraining, layerName + "_BN", st))
    if usedDropOut:
        x = _dropout(x, keepProb, isTraining)
    return x

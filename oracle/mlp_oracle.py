"""CPU oracle of the SDF network (TEST INFRASTRUCTURE -- checker only; never imported by gshell_amd/).

Restates the reference's geometry/mlp.py:7-40 (`MLP`: Linear + Softplus(beta=100) stack, the positional encoding concatenated to the
input of the layers listed in `skip_in`) and geometry/embedding.py:4-39 (`Embedding`: (x, sin(2^k x), cos(2^k x))_{k < n_freq}) as
plain functions of a state dict with the reference's key names (`net.{2i}.weight / bias`), in whatever dtype the state dict carries
-- float32 for the parity chains, float64 for the arbiter runs of oracle/make_golden_chain.py.

PINNED: tests/test_chain_recipe_cpu.py compares `forward` bit for bit with the REAL reference module (loaded from /root/reference when it
is present) and with the product's torch module on the CPU."""
import torch


def embed(x, n_freq=6):
    """embedding.py:22-39: the frequency loop multiplies by a python float taken from 2 ** linspace(0, n_freq - 1, n_freq)."""
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def softplus100(x):
    """nn.Softplus(beta=100) (mlp.py:13,24): log1p(exp(100 x)) / 100 with torch's linear branch above 100 x > 20."""
    return torch.nn.functional.softplus(x, beta=100.0, threshold=20.0)


def layer_keys(state):
    ids = sorted({int(k.split('.')[1]) for k in state if k.startswith('net.')})
    return ids


def forward(state, x, n_freq=6, skip_in=(3,)):
    """state: {'net.0.weight': [256,39], 'net.0.bias': [256], 'net.2.weight': ...}; x [n,3] -> [n,1].
    Hidden layer i (0-based, after the input layer) takes cat(h, PE) when i is in skip_in (mlp.py:19-21,:35-36)."""
    emb = embed(x, n_freq)
    ids = layer_keys(state)
    h = emb
    for j, lid in enumerate(ids):
        w, b = state[f'net.{lid}.weight'], state[f'net.{lid}.bias']
        hidden_index = j - 1                                   # layer 0 is the input layer
        if hidden_index in skip_in and j + 1 < len(ids):
            h = torch.cat([h, emb], -1)
        h = torch.nn.functional.linear(h, w, b)
        if j + 1 < len(ids):
            h = softplus100(h)
    return h


def forward_chunked(state, x, chunk=131072, **kw):
    """no-graph evaluation over many rows in bounded memory (values identical to `forward` row by row up to the GEMM's blocking)"""
    with torch.no_grad():
        return torch.cat([forward(state, x[i:i + chunk], **kw) for i in range(0, x.shape[0], chunk)], 0)

"""Cross-frame transformer decoder of branch B (reference: libs/models/utils/transformer.py:92-129, 236-298;
pre-norm layers, GELU feed-forward, nn.MultiheadAttention parameter layout)."""
import copy
import math

import torch
import torch.nn as nn

from phnet_amd import functional as PF


def _attention(mha: nn.MultiheadAttention, q_in: torch.Tensor, kv_in: torch.Tensor, training: bool, key_valid=None,
               batch: int = 1) -> torch.Tensor:
    """q_in [B*L,E], kv_in [B*M,E] (B clips as contiguous row blocks) -> [B*L,E]; key_valid bool[B*M] masks padded memory slots."""
    e, h = mha.embed_dim, mha.num_heads
    w, b = mha.in_proj_weight, mha.in_proj_bias
    p_drop = mha.dropout if training else 0.0
    if q_in is kv_in:
        qkv = PF.linear(q_in, w, b)                       # self-attention: one [L,3E] projection on the whole in_proj
        out = PF.attention_packed(qkv, h, p_drop, batch)
    else:
        q = PF.linear(q_in, w, b, rows=(0, e))
        kv = PF.linear(kv_in, w, b, rows=(e, 3 * e))
        out = PF.attention_cross(q, kv, h, p_drop, key_valid, batch)
    return PF.linear(out, mha.out_proj.weight, mha.out_proj.bias)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if not normalize_before or activation != "gelu":
            raise NotImplementedError("the hot path uses pre-norm GELU layers (Router4OL.py:100-103)")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, memory_key_valid=None, h=None, next_norm=None, batch: int = 1):
        """tgt [L,E], memory [M,E], memory_key_valid bool[M] or None.  h = norm1(tgt) if the caller already has it.
        Every residual update is fused with the LayerNorm that follows it (norm2, norm3, and `next_norm` = the next
        layer's norm1 or the decoder's final norm).  Returns (tgt_out, next_norm(tgt_out) or None)."""
        p = lambda d: d.p if self.training else 0.0                                  # noqa: E731
        if h is None:
            h = PF.layer_norm(tgt, self.norm1.weight, self.norm1.bias, eps=self.norm1.eps)
        tgt, h = PF.dropout_add_ln(tgt, _attention(self.self_attn, h, h, self.training, batch=batch),
                                   self.norm2.weight, self.norm2.bias, p(self.dropout1), self.norm2.eps)
        tgt, h = PF.dropout_add_ln(tgt, _attention(self.multihead_attn, h, memory, self.training, memory_key_valid, batch),
                                   self.norm3.weight, self.norm3.bias, p(self.dropout2), self.norm3.eps)
        h = PF.gelu_dropout(PF.linear(h, self.linear1.weight, self.linear1.bias), p(self.dropout))
        f = PF.linear(h, self.linear2.weight, self.linear2.bias)
        if next_norm is None:
            return PF.dropout_add(tgt, f, p(self.dropout3)), None
        return PF.dropout_add_ln(tgt, f, next_norm.weight, next_norm.bias, p(self.dropout3), next_norm.eps)


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = nn.ModuleList(copy.deepcopy(decoder_layer) for _ in range(num_layers))
        self.num_layers = num_layers
        self.norm = copy.deepcopy(norm)
        self.return_intermediate = return_intermediate

    fuse_row_chains = True      # no-autograd forward: the row-local chains between the attention cores as single launches

    def _forward_row_chains(self, x: torch.Tensor, mem: torch.Tensor, memory_key_valid, batch: int) -> torch.Tensor:
        """The same decoder with every row-local chain of a layer in one launch (csrc/rowchain.hip): per layer
        [LayerNorm + q|k|v projection] -> self-attention core -> [out-projection + residual/dropout + LayerNorm + q projection]
        -> cross-attention core -> [out-projection + residual/dropout + LayerNorm + feed-forward + residual/dropout + the next
        LayerNorm (+ the next layer's q|k|v projection)]: 5 launches + the memory's k|v projection instead of 13.  Forward only;
        dropout sites are drawn in the order of `TransformerDecoderLayer.forward`, so a recomputation through the unfused
        kernels sees the same masks."""
        from phnet_amd import hip_ops as K
        site = PF.DropoutStream.site
        dev = x.device
        x = x.contiguous()
        first = self.layers[0]
        e = x.shape[1]
        _, _, qkv = K.rowchain_fwd(x, ln1=(first.norm1.weight, first.norm1.bias), wg=first.self_attn.in_proj_weight,
                                   bg=first.self_attn.in_proj_bias, eps=first.norm1.eps, want_t=False)
        h = None
        for i, layer in enumerate(self.layers):
            tr = layer.training
            p_of = lambda d: d.p if tr else 0.0                                           # noqa: E731
            sa, ca = layer.self_attn, layer.multihead_attn
            a, _ = K.attention_fwd(qkv[:, :e], qkv[:, e:2 * e], qkv[:, 2 * e:], sa.num_heads, None,
                                   rng=site(dev, sa.dropout if tr else 0.0), batch=batch)
            t, _, q = K.rowchain_fwd(a, resid=x, wa=sa.out_proj.weight, ba=sa.out_proj.bias, ln1=(layer.norm2.weight, layer.norm2.bias),
                                     wg=ca.in_proj_weight[:e], bg=ca.in_proj_bias[:e], eps=layer.norm2.eps, rng_a=site(dev, p_of(layer.dropout1)))
            kv = K.linear_fwd(mem, ca.in_proj_weight[e:], ca.in_proj_bias[e:])
            kvu8 = None if memory_key_valid is None else memory_key_valid.contiguous().view(torch.uint8)
            a2, _ = K.attention_fwd(q, kv[:, :e], kv[:, e:], ca.num_heads, kvu8, rng=site(dev, ca.dropout if tr else 0.0), batch=batch)
            nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
            norm = nxt.norm1 if nxt is not None else self.norm
            r_a, r_f, r_3 = site(dev, p_of(layer.dropout2)), site(dev, p_of(layer.dropout)), site(dev, p_of(layer.dropout3))
            x, h, qkv = K.rowchain_fwd(a2, resid=t, wa=ca.out_proj.weight, ba=ca.out_proj.bias, ln1=(layer.norm3.weight, layer.norm3.bias),
                                       ffn=(layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias),
                                       ln2=(norm.weight, norm.bias) if norm is not None else (layer.norm3.weight, layer.norm3.bias),
                                       wg=None if nxt is None else nxt.self_attn.in_proj_weight,
                                       bg=None if nxt is None else nxt.self_attn.in_proj_bias, eps=layer.norm3.eps,
                                       rng_a=r_a, rng_f=r_f, rng_3=r_3, want_t=True, want_h=nxt is None)
        return h if self.norm is not None else x

    def _can_fuse(self, x, mem) -> bool:
        l0 = self.layers[0]
        e, ff = x.shape[-1], l0.linear1.out_features
        ps = {l.dropout.p for l in self.layers} | {l.dropout1.p for l in self.layers} | {l.dropout2.p for l in self.layers} | {l.dropout3.p for l in self.layers}
        return (self.fuse_row_chains and not torch.is_grad_enabled() and x.is_cuda and (e, ff) in ((128, 256), (256, 512))
                and l0.self_attn.embed_dim // l0.self_attn.num_heads in (16, 32) and len(ps) == 1 and self.norm is not None
                and (e == 128 or not self.training))

    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, memory_key_valid=None, batch: int = 1) -> torch.Tensor:
        """tgt [L,1,E] or [L,E]; memory [M,1,E] or [M,E]; memory_key_valid bool[M]; returns the same rank as tgt.
        batch = B: tgt [B*L,E], memory [B*M,E], mask [B*M] hold B clips as contiguous row blocks."""
        shape = tgt.shape
        x, mem = tgt.reshape(-1, shape[-1]), memory.reshape(-1, memory.shape[-1])
        if self._can_fuse(x, mem):
            return self._forward_row_chains(x, mem.contiguous(), memory_key_valid, batch).reshape(shape)
        h = None
        for i, layer in enumerate(self.layers):
            nxt = self.layers[i + 1].norm1 if i + 1 < len(self.layers) else self.norm
            x, h = layer(x, mem, memory_key_valid, h=h, next_norm=nxt, batch=batch)
        return (h if self.norm is not None else x).reshape(shape)

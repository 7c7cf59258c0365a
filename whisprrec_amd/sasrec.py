"""SASRec with its item-embedding gather / scatter on the HIP kernels (reference src/models/sequential/SASRec.py).

In scope for this path: ``item_embedding(padding_idx=0)`` looked up for the history, the positive
and the negative items (SASRec.py:84,105-106) and the scatter-add of its backward with the padding row's gradient
dropped.  ``HipEmbedding`` does exactly that through ``wr_gather_rows`` / ``wr_scatter_add_rows`` (sorted, deterministic).
The single transformer block (src/utils/layers.py:8-86) is dense T<=20 attention and stays on stock PyTorch-ROCm ops; it
is restated here with the reference's parameter names so checkpoints interchange.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip_ops, host


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, idx, padding_idx):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.padding_idx = weight.shape, padding_idx
        return hip_ops.gather_rows(weight.detach(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad = torch.zeros(ctx.shape, dtype=torch.float32, device=grad_out.device)
        hip_ops.scatter_add_rows(grad, idx, grad_out.contiguous(), padding_idx=ctx.padding_idx)
        return grad, None, None


class HipEmbedding(nn.Module):
    """nn.Embedding replacement: same ``weight`` parameter name/shape; ``padding_idx`` rows receive no gradient (their
    stored values are returned as they are — xavier init overwrites the zero row in the reference too)."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx = -1 if padding_idx is None else int(padding_idx)
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)
        if padding_idx is not None:
            with torch.no_grad():
                self.weight[padding_idx].fill_(0)

    def forward(self, idx):
        return _GatherRows.apply(self.weight, idx, self.padding_idx)


class _Attention(nn.Module):
    """Multi-head self-attention of layers.py:8-57: separate q/k/v projections, no output projection, scores shifted by
    their GLOBAL maximum before the softmax (:54), NaN rows zeroed (:55)."""

    def __init__(self, d_model, n_heads):
        super().__init__()
        self.h, self.d_k = n_heads, d_model // n_heads
        self.q_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)

    def _split(self, x):
        return x.view(*x.shape[:-1], self.h, self.d_k).transpose(-2, -3)

    def forward(self, x, mask):
        q, k, v = self._split(self.q_linear(x)), self._split(self.k_linear(x)), self._split(self.v_linear(x))
        s = torch.matmul(q, k.transpose(-2, -1)) / self.d_k ** 0.5
        s = s.masked_fill(mask == 0, -np.inf)
        s = (s - s.max()).softmax(dim=-1)
        s = s.masked_fill(torch.isnan(s), 0)
        return torch.matmul(s, v).transpose(-2, -3).reshape(x.shape)


class _Block(nn.Module):
    """layers.py:60-86: post-norm residual attention + 2-layer ReLU feed-forward."""

    def __init__(self, d_model, d_ff, n_heads, dropout):
        super().__init__()
        self.masked_attn_head = _Attention(d_model, n_heads)
        self.layer_norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.layer_norm2 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, seq, mask):
        ctx = self.layer_norm1(self.dropout1(self.masked_attn_head(seq, mask)) + seq)
        out = self.linear2(self.linear1(ctx).relu())
        return self.layer_norm2(self.dropout2(out) + ctx)


def _xavier_normal_all(module):
    """reference src/models/init.py:13-29 applied through nn.Module.apply"""
    if isinstance(module, (nn.Embedding, HipEmbedding)):
        nn.init.xavier_normal_(module.weight.data)
    elif isinstance(module, nn.Linear):
        nn.init.xavier_normal_(module.weight.data)
        if module.bias is not None:
            nn.init.constant_(module.bias.data, 0)


def make_sasrec(sequential_model_cls):
    class SASRec(sequential_model_cls):
        reader = "SeqReader"
        runner = "BaseRunner"
        extra_log_args = ["emb_size", "num_layers", "num_heads"]

        @staticmethod
        def parse_model_args(parser):
            parser.add_argument("--emb_size", type=int, default=64, help="Size of embedding vectors.")
            parser.add_argument("--num_layers", type=int, default=1, help="Number of self-attention layers.")
            parser.add_argument("--num_heads", type=int, default=4, help="Number of attention heads.")
            parser.add_argument("--dropout", type=float, default=0.1, help="Dropout probability.")
            return sequential_model_cls.parse_model_args(parser)

        def __init__(self, args, corpus):
            super().__init__(args, corpus)
            self.emb_size, self.max_his = args.emb_size, args.history_max
            self.item_embedding = HipEmbedding(self.item_num, self.emb_size, padding_idx=0)
            self.position_embedding = nn.Embedding(self.max_his + 1, self.emb_size)
            self.transformer_block = nn.ModuleList([_Block(self.emb_size, self.emb_size, args.num_heads, args.dropout)
                                                    for _ in range(args.num_layers)])
            self.apply(_xavier_normal_all)

        def forward(self, feed_dict):
            history, lengths = feed_dict["history_items"], feed_dict["lengths"]
            bsz, T = history.shape
            valid = (history > 0).float()
            pos_ids = torch.arange(T, device=history.device).unsqueeze(0).expand_as(history)
            x = self.item_embedding(history) + self.position_embedding(pos_ids)        # SASRec.py:79-85
            mask = torch.tril(torch.ones(1, 1, T, T, dtype=torch.int32, device=history.device))
            for blk in self.transformer_block:
                x = blk(x, mask)
            x = x * valid[:, :, None]
            return x[torch.arange(bsz, device=history.device), lengths - 1, :]          # last valid position (:95)

        def predict(self, feed_dict):
            user_e = self.forward(feed_dict)
            pos_e = self.item_embedding(feed_dict["pos_item"])
            neg_e = self.item_embedding(feed_dict["neg_items"].reshape(-1))
            pos = (user_e * pos_e).sum(dim=1)
            neg = (user_e * neg_e).sum(dim=1)
            return -torch.log(1e-10 + torch.sigmoid(pos - neg)).mean()                   # BPRLoss, loss.py:38

        def full_predict(self, feed_dict):
            return torch.matmul(self.forward(feed_dict), self.item_embedding.weight.t())

    SASRec.__qualname__ = "SASRec"
    return SASRec


SASRec = make_sasrec(host.SequentialModel)


def bind(reference_sequential_model_cls):
    return make_sasrec(reference_sequential_model_cls)

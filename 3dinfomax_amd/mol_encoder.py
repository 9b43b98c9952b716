"""Atom / bond feature encoders - drop-in for reference commons/mol_encoder.py (same module names:
`atom_embedding_list.{k}`, `bond_embedding_list.{k}`; xavier_uniform init, :26-27, 58-59).  The nn.Embedding
modules only hold the tables; the lookup-and-sum is one fused HIP kernel (csrc/embedding.hip)."""
from math import sqrt

import torch

from . import tape
from .layers import EmbeddingSumFn, EmbeddingSumPadFn
from .synth import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS

# ogb.utils.features.get_atom_feature_dims()/get_bond_feature_dims() (ogb >= 1.3); used from ogb when installed
try:  # pragma: no cover - ogb is absent in the build image
    from ogb.utils.features import get_atom_feature_dims, get_bond_feature_dims
    full_atom_feature_dims = get_atom_feature_dims()
    full_bond_feature_dims = get_bond_feature_dims()
except Exception:
    full_atom_feature_dims = list(ATOM_FEATURE_DIMS)
    full_bond_feature_dims = list(BOND_FEATURE_DIMS)


class _Encoder(torch.nn.Module):
    _list_name = None

    def _build(self, dims, emb_dim, padding):
        self.padding = padding
        lst = torch.nn.ModuleList()
        for dim in dims:
            # reference :22-27: one extra row with padding_idx=0, then xavier over the WHOLE table (row 0 included)
            emb = torch.nn.Embedding(dim + 1, emb_dim, padding_idx=0) if padding else torch.nn.Embedding(dim, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            lst.append(emb)
        return lst

    @property
    def dims(self):
        return [emb.num_embeddings for emb in getattr(self, self._list_name)]

    def _tables(self):
        return [emb.weight for emb in getattr(self, self._list_name)]

    def forward(self, x, perm=None):
        return tape.apply(EmbeddingSumPadFn if self.padding else EmbeddingSumFn, x.contiguous(), perm, *self._tables())


class AtomEncoder(_Encoder):
    """reference commons/mol_encoder.py:10-42."""
    _list_name = 'atom_embedding_list'

    def __init__(self, emb_dim, padding=False):
        super().__init__()
        self.atom_embedding_list = self._build(full_atom_feature_dims, emb_dim, padding)

    def reset_parameters(self):
        for embedder in self.atom_embedding_list:
            embedder.weight.data.uniform_(-sqrt(3), sqrt(3))


class BondEncoder(_Encoder):
    """reference commons/mol_encoder.py:45-73."""
    _list_name = 'bond_embedding_list'

    def __init__(self, emb_dim, padding=False):
        super().__init__()
        self.bond_embedding_list = self._build(full_bond_feature_dims, emb_dim, padding)

"""3dinfomax_amd - MI355X-native (gfx950, HIP) implementation of the 3DInfomax pre-training hot path:
PNA(+Net3D)+NT-Xent behind the reference's model_type / model3d_type / loss_func plugin surface.

The package name starts with a digit, so import it with importlib.import_module('3dinfomax_amd') or through
the alias module `infomax3d_amd` at the repository root.
"""
from .graph import (BatchedMolGraph, GraphIndex, as_batched_graph, batch, bond_graph, complete_graph,  # noqa: F401
                    conformer_collate, contrastive_collate, graph_collate, s_norm_contrastive_collate,
                    s_norm_graph_collate)
from . import synth  # noqa: F401


def __getattr__(name):
    # model / loss classes need torch + the HIP library: import lazily so that the numpy-only parts
    # (synth, graph) stay importable everywhere.
    if name in ('PNA', 'PNAGNN', 'PNALayer', 'PNA_AGGREGATORS', 'PNA_SCALERS'):
        from . import pna
        return getattr(pna, name)
    if name in ('PNAOriginal', 'PNAOriginalSimple', 'PNAGNNOriginal', 'PNAGNNSimple', 'PNATower', 'PNASimpleLayer',
                'MLPReadout'):
        from . import pna_original
        return getattr(pna_original, name)
    if name in ('Net3D', 'Net3DLayer'):
        from . import net3d
        return getattr(net3d, name)
    if name in ('NTXent', 'NTXentMultiplePositives'):
        from . import losses
        return getattr(losses, name)
    if name in ('FCLayer', 'MLP'):
        from . import layers
        return getattr(layers, name)
    if name in ('AtomEncoder', 'BondEncoder'):
        from . import mol_encoder
        return getattr(mol_encoder, name)
    if name in ('PositiveSimilarity', 'NegativeSimilarity', 'ContrastiveAccuracy', 'TrueNegativeRate', 'TruePositiveRate',
                'Uniformity', 'Alignment', 'BatchVariance', 'DimensionCovariance', 'contrastive_metrics'):
        from . import metrics
        return getattr(metrics, name)
    if name == 'Adam':
        from . import optim
        return optim.Adam
    if name in ('set_matmul_precision', 'get_matmul_precision', 'set_fp32_products', 'get_fp32_products'):
        from . import ops
        return getattr(ops, name)
    if name in ('dataset', 'dist', 'tape', 'streams', 'ops'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)


__all__ = ['PNA', 'PNAGNN', 'PNALayer', 'PNA_AGGREGATORS', 'PNA_SCALERS', 'PNAOriginal', 'PNAOriginalSimple',
           'PNAGNNOriginal', 'PNAGNNSimple', 'PNATower', 'PNASimpleLayer', 'MLPReadout', 'Net3D', 'Net3DLayer', 'NTXent',
           'NTXentMultiplePositives', 'FCLayer', 'MLP', 'AtomEncoder', 'BondEncoder', 'contrastive_collate',
           'conformer_collate', 'graph_collate', 's_norm_graph_collate', 's_norm_contrastive_collate', 'BatchedMolGraph', 'batch', 'bond_graph', 'complete_graph', 'Adam', 'PositiveSimilarity',
           'NegativeSimilarity', 'ContrastiveAccuracy', 'TrueNegativeRate', 'TruePositiveRate', 'Uniformity', 'Alignment',
           'BatchVariance', 'DimensionCovariance']

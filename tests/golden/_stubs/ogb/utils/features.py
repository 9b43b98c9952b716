"""ogb.utils.features constants (ogb >= 1.3): see package docstring."""


def get_atom_feature_dims():
    return [119, 5, 12, 12, 10, 6, 6, 2, 2]


def get_bond_feature_dims():
    return [5, 6, 2]

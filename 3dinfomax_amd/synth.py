"""Seeded synthetic molecule generators (QM9-shaped and QMugs-shaped).

The reference's data blobs (QM9 csv/npz, QMugs SDFs) are absent, so every benchmark,
golden vector and parity test runs on synthetic molecules that follow the tensor layout
the reference's datasets produce:

* bond graph: each bond stored as two directed edges (i,j),(j,i), adjacent, in bond order
  (reference datasets/qm9_dataset.py:431-435); atom features int64 [n,9] within the ogb
  atom feature dims, column 0 = atomic number - 1; bond features int64 [e,3] identical for
  both directions (reference datasets/qm9_dataset.py:425-437)
* complete graph without self loops, edge order src = repeat_interleave(arange(n), n-1),
  dst = all j != src ascending (reference datasets/qm9_dataset.py:215-217), edge datum
  d = Euclidean distance [E3,1] (reference datasets/qm9_dataset.py:241-242)

numpy only - no torch, no HIP - so the oracle, the golden generator and the product path
all share the same inputs.
"""
from dataclasses import dataclass

import numpy as np

# ogb.utils.features.get_atom_feature_dims() / get_bond_feature_dims() for ogb >= 1.3
# (third-party constant used by reference commons/mol_encoder.py:6-7).
ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


@dataclass
class Molecule:
    n_atoms: int
    src: np.ndarray         # int64 [e]   directed bond edges, edge-id order
    dst: np.ndarray         # int64 [e]
    atom_feat: np.ndarray   # int64 [n,9]
    bond_feat: np.ndarray   # int64 [e,3]
    coords: np.ndarray      # float32 [n,3]


# element tables: (atomic number, valence cap, probability)
_QM9_ELEMENTS = [(6, 4, 0.72), (7, 3, 0.12), (8, 2, 0.15), (9, 1, 0.01)]
_QMUGS_ELEMENTS = [(6, 4, 0.62), (7, 3, 0.12), (8, 2, 0.14), (16, 6, 0.04), (9, 1, 0.03),
                   (17, 1, 0.03), (15, 5, 0.02)]


def _build_molecule(rng, n_heavy, elements, h_prob, max_rings, coord_sigma):
    z_tab = np.array([e[0] for e in elements])
    cap_tab = np.array([e[1] for e in elements])
    p_tab = np.array([e[2] for e in elements], dtype=np.float64)
    p_tab /= p_tab.sum()
    kind = rng.choice(len(elements), size=n_heavy, p=p_tab)
    # the first atom must be able to carry a tree: force a multivalent element
    if cap_tab[kind[0]] < 2:
        kind[0] = 0
    z = z_tab[kind]
    free = cap_tab[kind].copy()
    bonds = []
    for i in range(1, n_heavy):
        cand = np.nonzero(free[:i] > 0)[0]
        if cand.size == 0:       # nothing left to attach to: re-type a previous atom as carbon
            j = int(rng.integers(0, i))
            free[j] += 1
        else:
            j = int(cand[rng.integers(0, cand.size)])
        if free[i] == 0:
            free[i] = 1
        bonds.append((j, i))
        free[j] -= 1
        free[i] -= 1
    adj = {(a, b) for a, b in bonds} | {(b, a) for a, b in bonds}
    for _ in range(int(rng.integers(0, max_rings + 1))):
        cand = np.nonzero(free > 0)[0]
        if cand.size < 2:
            break
        a, b = rng.choice(cand, size=2, replace=False)
        a, b = int(a), int(b)
        if (a, b) in adj:
            continue
        bonds.append((a, b))
        adj.add((a, b))
        adj.add((b, a))
        free[a] -= 1
        free[b] -= 1
    # hydrogens: each remaining valence becomes an H with probability h_prob (the rest is
    # thought of as absorbed by multiple bonds - bond order does not change the topology)
    zs = list(z)
    n = n_heavy
    for a in range(n_heavy):
        for _ in range(int(free[a])):
            if rng.random() < h_prob:
                bonds.append((a, n))
                zs.append(1)
                n += 1
    rng.shuffle(bonds)  # bond order in an SDF is not sorted by atom index
    src = np.empty(2 * len(bonds), dtype=np.int64)
    dst = np.empty(2 * len(bonds), dtype=np.int64)
    for k, (a, b) in enumerate(bonds):
        src[2 * k], dst[2 * k] = a, b
        src[2 * k + 1], dst[2 * k + 1] = b, a
    atom_feat = np.stack([rng.integers(0, d, size=n) for d in ATOM_FEATURE_DIMS], axis=1).astype(np.int64)
    atom_feat[:, 0] = np.array(zs) - 1
    bf = np.stack([rng.integers(0, d, size=len(bonds)) for d in BOND_FEATURE_DIMS], axis=1).astype(np.int64)
    bond_feat = np.repeat(bf, 2, axis=0)
    coords = rng.normal(0.0, coord_sigma, size=(n, 3)).astype(np.float32)
    return Molecule(n, src, dst, atom_feat, bond_feat, coords)


def qm9_like(rng) -> Molecule:
    """One QM9-shaped molecule: 9 heavy atoms (85 %), ~18 atoms, degrees 1..4."""
    r = rng.random()
    n_heavy = 9 if r < 0.85 else (8 if r < 0.96 else int(rng.integers(4, 8)))
    return _build_molecule(rng, n_heavy, _QM9_ELEMENTS, h_prob=0.55, max_rings=2, coord_sigma=1.5)


def qmugs_like(rng) -> Molecule:
    """One QMugs-shaped molecule: n ~ clip(N(55,18), 8, 200) atoms, degrees <= 6."""
    n_target = int(np.clip(rng.normal(55, 18), 8, 200))
    n_heavy = max(4, int(round(n_target * 0.52)))
    return _build_molecule(rng, n_heavy, _QMUGS_ELEMENTS, h_prob=0.5, max_rings=4, coord_sigma=3.0)


def make_dataset(n_molecules, seed=0, kind='qm9'):
    """List of `n_molecules` seeded molecules."""
    rng = np.random.default_rng(seed)
    gen = qm9_like if kind == 'qm9' else qmugs_like
    return [gen(rng) for _ in range(n_molecules)]


def conformers(mol: Molecule, rng, num_conformers=3, noise=0.05):
    """Coordinate sets of `num_conformers` conformers: coords + N(0, noise)
    (cf. reference datasets/qmugs_dataset.py:158-159)."""
    return [mol.coords + rng.normal(0, noise, size=mol.coords.shape).astype(np.float32)
            for _ in range(num_conformers)]


def complete_graph_edges(n):
    """(src, dst) of the complete graph without self loops in the reference's order
    (reference datasets/qm9_dataset.py:215-217)."""
    ar = np.arange(n, dtype=np.int64)
    src = np.repeat(ar, n - 1)
    dst = np.concatenate([np.concatenate([ar[:i], ar[i + 1:]]) for i in range(n)]) if n > 1 else ar[:0]
    return src, dst


def pairwise_distances(coords, src, dst):
    """edata['d'] of the complete graph: ||x_src - x_dst||_2, shape [E3,1] float32
    (reference datasets/qm9_dataset.py:241-242)."""
    diff = coords[src] - coords[dst]
    return np.sqrt((diff.astype(np.float32) ** 2).sum(-1, dtype=np.float32))[:, None].astype(np.float32)

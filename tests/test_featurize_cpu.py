"""SMILES -> graph featuriser (gnnepcsaft_amd/data/featurize.py; stands in for ogb's ``smiles2graph`` on RDKit,
/root/reference/gnnepcsaft/data/ogb_utils.py:37-147).  PARITY UNPINNED: RDKit is not installable here, so the expected
vectors below are known answers for common molecules (what RDKit + ogb 1.3.6 produce for them, from the published
feature definitions), plus structural invariants on a committed sample of the reference's own molecule table."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# node: [Z-1, chirality, total degree, charge+5, #H, radicals, hybridisation(SP,SP2,SP3,..), aromatic, in ring]
# edge: [type (single, double, triple, aromatic), stereo (none, Z, E), conjugated]
KNOWN = {
    "CCO": ([[5, 0, 4, 5, 3, 0, 2, 0, 0], [5, 0, 4, 5, 2, 0, 2, 0, 0], [7, 0, 2, 5, 1, 0, 2, 0, 0]], [[0, 0, 0], [0, 0, 0]]),
    "O": ([[7, 0, 2, 5, 2, 0, 2, 0, 0]], []),
    "[NH4+]": ([[6, 0, 4, 6, 4, 0, 2, 0, 0]], []),
    "C#N": ([[5, 0, 2, 5, 1, 0, 0, 0, 0], [6, 0, 1, 5, 0, 0, 0, 0, 0]], [[2, 0, 0]]),
    "CC(=O)O": ([[5, 0, 4, 5, 3, 0, 2, 0, 0], [5, 0, 3, 5, 0, 0, 1, 0, 0], [7, 0, 1, 5, 0, 0, 1, 0, 0],
                 [7, 0, 2, 5, 1, 0, 1, 0, 0]], [[0, 0, 0], [1, 0, 1], [0, 0, 1]]),  # hydroxyl O is SP2: conjugated
    "CC(=O)N": ([[5, 0, 4, 5, 3, 0, 2, 0, 0], [5, 0, 3, 5, 0, 0, 1, 0, 0], [7, 0, 1, 5, 0, 0, 1, 0, 0],
                 [6, 0, 3, 5, 2, 0, 1, 0, 0]], [[0, 0, 0], [1, 0, 1], [0, 0, 1]]),
    "C=CC=C": ([[5, 0, 3, 5, 2, 0, 1, 0, 0], [5, 0, 3, 5, 1, 0, 1, 0, 0], [5, 0, 3, 5, 1, 0, 1, 0, 0],
                [5, 0, 3, 5, 2, 0, 1, 0, 0]], [[1, 0, 1], [0, 0, 1], [1, 0, 1]]),
    "c1ccncc1": ([[5, 0, 3, 5, 1, 0, 1, 1, 1]] * 3 + [[6, 0, 2, 5, 0, 0, 1, 1, 1]] + [[5, 0, 3, 5, 1, 0, 1, 1, 1]] * 2,
                 [[3, 0, 1]] * 6),
    "c1cc[nH]c1": ([[5, 0, 3, 5, 1, 0, 1, 1, 1]] * 3 + [[6, 0, 3, 5, 1, 0, 1, 1, 1]] + [[5, 0, 3, 5, 1, 0, 1, 1, 1]],
                   [[3, 0, 1]] * 5),
    "C1CCCCC1": ([[5, 0, 4, 5, 2, 0, 2, 0, 1]] * 6, [[0, 0, 0]] * 6),
    "CS(=O)(=O)C": ([[5, 0, 4, 5, 3, 0, 2, 0, 0], [15, 0, 4, 5, 0, 0, 2, 0, 0], [7, 0, 1, 5, 0, 0, 1, 0, 0],
                     [7, 0, 1, 5, 0, 0, 1, 0, 0], [5, 0, 4, 5, 3, 0, 2, 0, 0]], [[0, 0, 0], [1, 0, 0], [1, 0, 0], [0, 0, 0]]),
    "F/C=C/F": ([[8, 0, 1, 5, 0, 0, 2, 0, 0], [5, 0, 3, 5, 1, 0, 1, 0, 0], [5, 0, 3, 5, 1, 0, 1, 0, 0],
                 [8, 0, 1, 5, 0, 0, 2, 0, 0]], [[0, 0, 0], [1, 2, 0], [0, 0, 0]]),   # E
    "F/C=C\\F": ([[8, 0, 1, 5, 0, 0, 2, 0, 0], [5, 0, 3, 5, 1, 0, 1, 0, 0], [5, 0, 3, 5, 1, 0, 1, 0, 0],
                  [8, 0, 1, 5, 0, 0, 2, 0, 0]], [[0, 0, 0], [1, 1, 0], [0, 0, 0]]),  # Z
}


@pytest.mark.parametrize("smiles", list(KNOWN))
def test_known_answers(smiles):
    from gnnepcsaft_amd.data.featurize import smiles2graph
    g = smiles2graph(smiles)
    x, e = KNOWN[smiles]
    assert g["node_feat"].tolist() == x
    assert g["edge_feat"][0::2].tolist() == e and g["edge_feat"][1::2].tolist() == e  # both directions, same features
    assert g["node_feat"].dtype == np.int64 and g["edge_index"].dtype == np.int64 and g["edge_feat"].dtype == np.int64
    assert g["edge_index"].shape == (2, 2 * len(e)) and g["edge_feat"].shape == (2 * len(e), 3)
    if e:
        assert (g["edge_index"][:, 0::2] == g["edge_index"][::-1, 1::2]).all()  # (i, j) then (j, i)


def test_kekule_and_aromatic_input_agree():
    from gnnepcsaft_amd.data.featurize import smiles2graph
    for kek, aro in [("C1=CC=CC=C1", "c1ccccc1"), ("C1=CC=C2C=CC=CC2=C1", "c1ccc2ccccc2c1"), ("C1=COC=C1", "c1cocc1"),
                     ("CC1=CC=CC=C1", "Cc1ccccc1"), ("C1=CC=NC=C1", "c1ccncc1"), ("OC1=CC=CC=C1", "Oc1ccccc1")]:
        a, b = smiles2graph(kek), smiles2graph(aro)
        assert sorted(a["node_feat"].tolist()) == sorted(b["node_feat"].tolist()), (kek, aro)
        assert sorted(a["edge_feat"].tolist()) == sorted(b["edge_feat"].tolist()), (kek, aro)
    assert smiles2graph("C1=CCCC=C1")["node_feat"][:, 7].sum() == 0       # cyclohexadiene: 4 pi electrons, not aromatic
    assert smiles2graph("C1=CC=CC=CC=C1")["node_feat"][:, 7].sum() == 0   # cyclooctatetraene: 8
    assert smiles2graph("O=C1C=CC(=O)C=C1")["node_feat"][:, 7].sum() == 0  # quinone


def test_chirality_tags_and_their_loss_on_non_stereocentres():
    from gnnepcsaft_amd.data.featurize import smiles2graph
    assert smiles2graph("C[C@H](N)C(=O)O")["node_feat"][1, 1] == 2      # '@'  -> CHI_TETRAHEDRAL_CCW
    assert smiles2graph("C[C@@H](N)C(=O)O")["node_feat"][1, 1] == 1     # '@@' -> CHI_TETRAHEDRAL_CW
    assert smiles2graph("C[C@H](C)O")["node_feat"][1, 1] == 0           # two methyls: not a stereo centre
    a = smiles2graph("N[C@@H](C)C(=O)O")["node_feat"][1, 1]             # the same L-alanine written from the N
    assert a == 1


def test_single_atoms_errors_and_explicit_hydrogens():
    from gnnepcsaft_amd.data.featurize import smiles2graph
    g = smiles2graph("C")
    assert g["edge_index"].shape == (2, 0) and g["edge_feat"].shape == (0, 3) and g["node_feat"].tolist() == [[5, 0, 4, 5, 4, 0, 2, 0, 0]]
    assert smiles2graph("[H]C([H])([H])[H]")["node_feat"].tolist() == g["node_feat"].tolist()  # explicit H folded in
    assert smiles2graph("[Na+].[Cl-]")["num_nodes"] == 2 and smiles2graph("[Na+].[Cl-]")["edge_index"].shape == (2, 0)
    for bad in ("C1CC", "C(C", "Cx", ""):
        with pytest.raises(ValueError, match="SMILES is not valid"):
            smiles2graph(bad)


def _sample():
    rows = []
    for line in open(os.path.join(ROOT, "tests", "golden", "esper_smiles_sample.tsv")):
        if not line.startswith("#"):
            s, m, sig, eps = line.rstrip("\n").split("\t")
            rows.append((s, float(m), float(sig), float(eps)))
    return rows


def test_reference_molecules_feed_the_hot_path_containers():
    """94 molecules of the reference's Esper table: every feature inside its vocabulary (ogb_utils.py:8-33), graphs
    collate like PyG batches, calc_deg gives the PNA histogram, valence sanity (C: degree + unsaturation = 4)."""
    from gnnepcsaft_amd.data import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS, Batch, calc_deg
    from gnnepcsaft_amd.data.featurize import from_smiles
    rows = _sample()
    assert len(rows) == 94
    data = [from_smiles(s, para=torch.tensor([[m, sig, eps]], dtype=torch.float32)) for s, m, sig, eps in rows]
    for d in data:
        assert all(int(d.x[:, k].max()) < ATOM_FEATURE_DIMS[k] and int(d.x[:, k].min()) >= 0 for k in range(9)), d.smiles
        if d.edge_attr.numel():
            assert all(int(d.edge_attr[:, k].max()) < BOND_FEATURE_DIMS[k] for k in range(3)), d.smiles
            deg = torch.bincount(d.edge_index[1], minlength=d.num_nodes)
            carbons = d.x[:, 0] == 5
            assert bool((d.x[carbons, 2] <= 4).all()) and bool((deg <= d.x[:, 2]).all()), d.smiles
    batch = Batch.from_data_list(data)
    assert batch.x.shape[1] == 9 and batch.edge_attr.shape[1] == 3 and batch.num_graphs == 94
    assert batch.para.shape == (94, 3) and int(batch.edge_index.max()) < batch.x.size(0)
    deg = calc_deg(data)
    assert len(deg) <= 5 and sum(deg) == batch.x.size(0)  # heavy-atom in-degrees of organic molecules: 0..4

"""Input side of the hot path: graph containers, PyG-style collation, degree histogram, synthetic graphs."""
from .batching import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS, Batch, Data, calc_deg, in_degree, shard_by_graph
from .featurize import from_smiles, smiles2graph
from .synthetic import default_config, synthetic_batch

__all__ = ["ATOM_FEATURE_DIMS", "BOND_FEATURE_DIMS", "Batch", "Data", "calc_deg", "in_degree", "shard_by_graph",
           "default_config", "synthetic_batch", "from_smiles", "smiles2graph"]

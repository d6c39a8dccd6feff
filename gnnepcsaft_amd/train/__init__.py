"""Mirror of ``gnnepcsaft.train`` for the GNN hot path."""
from .models import GNNePCSAFT, GNNePCSAFTL, create_model, get_conv, get_global_pool  # noqa: F401
from .trainer import DataLoader, Trainer  # noqa: F401

"""CPU oracle for the gnnepcsaft GNN forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / reported CPU baseline.

PARITY UNPINNED.  The reference (`/root/reference/gnnepcsaft/train/models.py`)
only *wires* third-party modules (torch_geometric 2.x ``PNAConv``/``GINEConv``/
``BatchNorm``/``aggr.*``, ogb 1.3.6 ``AtomEncoder``/``BondEncoder``); neither
package is importable in this image and the reference ships no tests, golden
vectors or fixtures for this path.  This oracle is therefore
a restatement of the *published* algorithm of those packages, written with the
same ATen primitives PyG's CPU path bottoms out in (``index_select``, ``cat``,
``F.linear``, ``scatter_add_``, ``scatter_reduce_(amin/amax, include_self=False)``,
``F.batch_norm``, ``F.embedding``), pinned only by

* the torch-level known-answer facts listed in SURVEY.md §8c (ties, empty
  segments, std clamp edges, xavier bound) — ``tests/test_oracle_known_answers.py``;
* an independent per-node numpy loop restatement (``oracle/numpy_loops.py``);
* fp64 ``torch.autograd.gradcheck`` of every block.
"""

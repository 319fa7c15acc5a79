"""Empty stub (kge/util/subgraph.py:5)."""

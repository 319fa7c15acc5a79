"""Empty stub (kge/job/search_ax.py:3-8)."""
Models = None

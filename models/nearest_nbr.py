"""Drop-in for the flat-search API of the reference's models/nearest_nbr.py (the BERT embedding helpers are out of scope)."""
from anncur_amd.nearest_nbr import FlatIPIndex, build_flat_or_ivff_index  # noqa: F401

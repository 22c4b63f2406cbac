"""Drop-in subset of the reference's eval/eval_utils.py for the CUR path: compute_overlap (same output format).
The BERT embedding helpers of the reference module are out of scope (they need trained encoders)."""
from anncur_amd.eval_utils import compute_overlap, flatten_overlap, overlap_stats_from_counts  # noqa: F401

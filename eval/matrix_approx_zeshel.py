"""Drop-in for the reference module of the same name: `from eval.matrix_approx_zeshel import CURApprox` now returns the
MI355X operator (anncur_amd/cur.py).  plot_heat_map keeps the reference's call signature (matplotlib, not accelerated)."""
import logging
import sys

import numpy as np

from anncur_amd.cur import CURApprox  # noqa: F401  (same constructor / methods / errors as the reference class)

logging.basicConfig(stream=sys.stderr, format="%(asctime)s - %(levelname)s - %(name)s - %(message)s ",
					datefmt="%d/%m/%Y %H:%M:%S", level=logging.INFO)
LOGGER = logging.getLogger(__name__)


def plot_heat_map(val_matrix, row_vals, col_vals, metric, top_k, curr_res_dir, title=None, fname=None):
	"""Annotated heat map of one metric over the (anchor queries x anchor items) grid -> {curr_res_dir}/{fname or metric}.pdf."""
	import os
	import matplotlib
	matplotlib.use("Agg")
	import matplotlib.pyplot as plt
	val_matrix = np.asarray(val_matrix, dtype=np.float64)
	fig, ax = plt.subplots(figsize=(1.2 * len(col_vals) + 3, 0.8 * len(row_vals) + 2))
	im = ax.imshow(val_matrix, cmap="viridis", aspect="auto")
	ax.set_xticks(np.arange(len(col_vals))); ax.set_xticklabels([str(c) for c in col_vals])
	ax.set_yticks(np.arange(len(row_vals))); ax.set_yticklabels([str(r) for r in row_vals])
	ax.set_xlabel("Number of anchor entities"); ax.set_ylabel("Number of anchor mentions")
	for i in range(val_matrix.shape[0]):
		for j in range(val_matrix.shape[1]):
			ax.text(j, i, "{:.2f}".format(val_matrix[i, j]), ha="center", va="center", color="w")
	ax.set_title(title if title is not None else f"{metric} (top_k={top_k})")
	fig.colorbar(im, ax=ax)
	os.makedirs(curr_res_dir, exist_ok=True)
	out = os.path.join(curr_res_dir, f"{fname if fname is not None else metric}.pdf")
	fig.savefig(out, bbox_inches="tight")
	plt.close(fig)
	return out
